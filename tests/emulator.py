"""CPU emulation of the device op list (test infrastructure).

Interprets the ctypes op array a ``FlowEngine`` builds -- the very descriptors the HIP library
would receive -- with torch-CPU arithmetic, by mapping raw pointers back onto the engine's CPU
tensors.  This lets the `-m "not gpu"` suite check the engine's layout permutation, mask-aware
weight slicing, fusion and pointer arithmetic against the golden vectors without a GPU.
It implements the *documented semantics* of ``usf_linear_f32`` / ``usf_coupling_additive_f32``
(include/usflows_hip.h), not the kernels."""
import ctypes as C

import torch

from usflows_amd import _ext


class PtrMap:
    def __init__(self):
        self.tensors = []

    def add(self, t):
        if t is not None and torch.is_tensor(t) and t.numel() > 0:
            self.tensors.append(t)

    def view(self, ptr, rows, cols, ld, dtype=torch.float32):
        """strided [rows, cols] view (row stride ld) at raw address ptr"""
        for t in self.tensors:
            base, nbytes = t.data_ptr(), t.numel() * t.element_size()
            if base <= ptr < base + nbytes:
                off = (ptr - base) // t.element_size()
                flat = t.view(-1)
                assert off + (rows - 1) * ld + cols <= flat.numel(), "descriptor reads past its tensor"
                return torch.as_strided(flat, (rows, cols), (ld, 1), t.storage_offset() + off)
        raise KeyError(f"pointer {ptr:#x} not inside any known tensor")

    def vec(self, ptr, n):
        return self.view(ptr, 1, n, n)[0]


def emulate_linear(d, pm: PtrMap, dtype=torch.float32):
    M, N, K = d.M, d.N, d.K
    A = pm.view(d.A, M, K, d.lda).to(dtype)
    W = pm.view(d.W, N, K, d.ldw).to(dtype)
    if d.pre_div:
        A = A / pm.vec(d.pre_div, K).to(dtype)
    if d.pre_sub:
        A = A - pm.vec(d.pre_sub, K).to(dtype)
    v = A @ W.t()
    if d.bias:
        v = v + pm.vec(d.bias, N).to(dtype)
    if d.addend:
        v = v + pm.view(d.addend, M, N, d.ldadd).to(dtype)
    if d.act == _ext.ACT_LEAKY_RELU:
        v = torch.where(v > 0, v, v * d.slope)
    if d.residual:
        v = pm.view(d.residual, M, N, d.ldr).to(dtype) + d.res_sign * v
    if d.post_mul:
        v = v * pm.vec(d.post_mul, N).to(dtype)
    pm.view(d.C, M, N, d.ldc).copy_(v.to(torch.float32))


def emulate_gated_norm(d, pm: PtrMap, dtype=torch.float32):
    """usf_gated_norm_rows_f32 as include/usflows_hip.h documents it"""
    M, Cn, Cp = d.M, d.C, d.c_pad
    r = pm.view(d.skip, M, Cn, d.ld_skip).to(dtype)
    if d.vg:
        r = r + pm.view(d.vg, M, Cn, d.ld_vg).to(dtype) * torch.sigmoid(pm.view(d.vg + 4 * d.gate_off, M, Cn, d.ld_vg).to(dtype))
    if d.gamma:
        mean = r.mean(dim=1, keepdim=True)
        var = ((r - mean) ** 2).mean(dim=1, keepdim=True)
        r = (r - mean) / torch.sqrt(var + d.eps) * pm.vec(d.gamma, Cn).to(dtype) + pm.vec(d.beta, Cn).to(dtype)
    full = torch.zeros(M, Cp, dtype=dtype)
    full[:, :Cn] = r
    if d.out:
        pm.view(d.out, M, Cp, d.ld_out).copy_(full.to(torch.float32))
    if d.out_act:
        a = torch.where(full > 0, full, full * d.slope) if d.act == _ext.ACT_LEAKY_RELU else full
        pm.view(d.out_act, M, Cp, d.ld_act).copy_(a.to(torch.float32))


def emulate_coupling(d, pm: PtrMap, dtype=torch.float32):
    M = d.M
    gated = d.act == _ext.ACT_GATE

    def finish(v, l):
        """layer l's nonlinearity -- or, in gate mode, leaky_relu_backward from the saved activation -- and the optional side output"""
        if gated:
            gt = pm.view(d.gate[l], M, d.hidden[l], d.ld_gate).to(dtype)
            v = v * torch.where(gt > 0, torch.ones_like(v), torch.full_like(v, d.slope))
        elif d.act == _ext.ACT_LEAKY_RELU:
            v = torch.where(v > 0, v, v * d.slope)
        if d.hidden_out[l]:
            pm.view(d.hidden_out[l], M, d.hidden[l], d.ld_hidden_out).copy_(v.to(torch.float32))
        return v

    zp = pm.view(d.z + 4 * d.off_pass, M, d.n_pass, d.ldz).to(dtype)
    h = zp @ pm.view(d.W_in, d.hidden[0], d.n_pass, d.ldw_in).to(dtype).t() + pm.vec(d.b_in, d.hidden[0]).to(dtype)
    if d.context and not gated:
        ctx = pm.vec(d.context, M).to(dtype)
        h = h + (ctx[:, None] * pm.vec(d.W_ctx, d.hidden[0]).to(dtype)[None, :] + pm.vec(d.b_ctx, d.hidden[0]).to(dtype))
    h = finish(h, 0)
    for j in range(d.n_hidden - 1):
        W = pm.view(d.W_hid[j], d.hidden[j + 1], d.hidden[j], d.ldw_hid[j]).to(dtype)
        h = finish(h @ W.t() + pm.vec(d.b_hid[j], d.hidden[j + 1]).to(dtype), j + 1)
    Wo = pm.view(d.W_out, d.n_trans, d.hidden[d.n_hidden - 1], d.ldw_out).to(dtype)
    t = h @ Wo.t() + pm.vec(d.b_out, d.n_trans).to(dtype)
    zt = pm.view(d.z + 4 * d.off_trans, M, d.n_trans, d.ldz).to(dtype)
    pm.view(d.out + 4 * d.off_trans, M, d.n_trans, d.ldo).copy_((zt + d.sign * t).to(torch.float32))
    if d.out != d.z:
        pm.view(d.out + 4 * d.off_pass, M, d.n_pass, d.ldo).copy_(pm.view(d.z + 4 * d.off_pass, M, d.n_pass, d.ldz))


# ---- planes pipeline (include/usflows_hip.h: usf_pack_planes_f32 / usf_gemm_planes_bf16x3) -------------------------
def _slot_feature(s):
    return 16 * ((s & 7) >> 2) + 4 * (s >> 3) + (s & 3)


_SLOT_OF_FEATURE = [0] * 32
for _s in range(32):
    _SLOT_OF_FEATURE[_slot_feature(_s)] = _s


def planes_view(pm: PtrMap, ptr, npanels, nkb, fmt=0):
    """[npanels, nkb, NPL, 64 lines, 8] view of a planes buffer at raw address ptr (bf16 x 3 or fp16 x 2)"""
    npl, dt = (2, torch.float16) if fmt == 1 else (3, torch.bfloat16)
    raw = pm.view(ptr, 1, npanels * nkb * npl * 1024, npanels * nkb * npl * 1024, dtype=torch.uint8)[0]
    return raw.view(dt).view(npanels, nkb, npl, 64, 8)


def planes_decode(v, M):
    """logical fp32 matrix [M, 32 nkb] of a planes view (sum of the planes in fp32, exactly as the kernel reads it back)"""
    npan, nkb = v.shape[0], v.shape[1]
    x = v[:, :, 0].float() + v[:, :, 1].float()                                   # [npan, nkb, 64, 8]
    if v.shape[2] == 3:
        x = x + v[:, :, 2].float()
    x = x.view(npan, nkb, 4, 16, 8)                                               # line = 16 g + j -> [g, j, u]
    out = torch.zeros(npan, 16, nkb, 32)
    for g in range(4):
        for u in range(8):
            out[:, :, :, _slot_feature(8 * g + u)] = x[:, :, g, :, u].permute(0, 2, 1)
    return out.reshape(npan * 16, nkb * 32)[:M]


def planes_encode(v, X, kb0):
    """write logical fp32 matrix X [M, 32 nb] into blocks kb0 .. of a planes view (3-way round-to-nearest split)"""
    npan = v.shape[0]
    M, nb = X.shape[0], X.shape[1] // 32
    Xp = torch.zeros(npan * 16, nb * 32)
    Xp[:M] = X
    Xp = Xp.view(npan, 16, nb, 32)
    dt = v.dtype
    p1 = Xp.to(dt)
    r = Xp - p1.float()
    p2 = r.to(dt)
    planes = (p1, p2, (r - p2.float()).to(dt)) if v.shape[2] == 3 else (p1, p2)
    for pl, P in enumerate(planes):
        for g in range(4):
            for u in range(8):
                v[:, kb0: kb0 + nb, pl, 16 * g: 16 * g + 16, u] = P[:, :, :, _slot_feature(8 * g + u)].permute(0, 2, 1)


def emulate_pack_planes(d, pm: PtrMap, dtype=torch.float32):
    M, nkb = d.M, d.nkb
    npan = -(-M // 16)
    idx = pm.view(d.idx, 1, 32 * nkb, 32 * nkb, dtype=torch.int32)[0].long()
    src = pm.view(d.src, M, int(idx.max()) + 1, d.ld)
    X = torch.zeros(M, 32 * nkb)
    ok = idx >= 0
    X[:, ok] = src[:, idx[ok]]
    if d.pre_div:
        X[:, ok] = X[:, ok] / pm.vec(d.pre_div, 32 * nkb)[ok]
    if d.pre_sub:
        X[:, ok] = X[:, ok] - pm.vec(d.pre_sub, 32 * nkb)[ok]
    if d.format == 1 and d.range_flag and not bool((X.abs() < 65000.0).all()):
        pm.view(d.range_flag, 1, 1, 1, dtype=torch.int32)[0, 0] = 1
    planes_encode(planes_view(pm, d.planes, npan, nkb, d.format), X, 0)


def emulate_gemm_planes(d, pm: PtrMap, dtype=torch.float32):
    M = d.M
    npan = -(-M // 16)
    fmt = d.format
    npl = 2 if fmt == 1 else 3
    A = planes_decode(planes_view(pm, d.A, npan, d.a_nkb, fmt), M)[:, 32 * d.a_kb0: 32 * (d.a_kb0 + d.nk)]
    Wp = pm.view(d.W_planes, npl, d.w_rows * d.ldw, d.w_plane_stride).view(npl, d.w_rows, d.ldw)
    W = Wp[0].float() + Wp[1].float()
    if npl == 3:
        W = W + Wp[2].float()
    W = W[:, : 32 * d.nk]                                                             # K axis in physical slot order
    slot = torch.tensor([32 * (c // 32) + _SLOT_OF_FEATURE[c % 32] for c in range(32 * d.nk)])
    W = W[:, slot]                                                                    # -> logical order
    v = A.to(dtype) @ W.to(dtype).t()
    if d.bias:
        v = v + pm.vec(d.bias, d.w_rows).to(dtype)
    if d.act == _ext.ACT_LEAKY_RELU:
        v = torch.where(v > 0, v, v * d.slope)
    if d.C_f32 or d.base_part:
        if d.post_mul:
            v = v * pm.vec(d.post_mul, d.w_rows).to(dtype)
        if fmt == 1 and d.range_flag and not bool(torch.isfinite(v[:, : d.N]).all()):
            pm.view(d.range_flag, 1, 1, 1, dtype=torch.int32)[0, 0] = 1
        if d.base_part:        # base density in the epilogue: one partial sum per (row, column block of 32 TN columns)
            tab = pm.view(d.base_tab, 3, d.N, d.base_tab_stride).to(dtype)
            t = (v[:, : d.N] - tab[0]) * tab[1]
            term = tab[2] - (t.abs() if d.base == _ext.BASE_LAPLACE else 0.5 * t * t)
            bn = 32 * ((_ext.load().usf_gemm_planes_variant(d) - 5000) // 10)
            part = pm.view(d.base_part, M, 8, 8)
            part.fill_(float("nan"))                     # (slots beyond the column blocks are never written by the kernel)
            for j in range(-(-d.N // bn)):
                part[:, j] = term[:, j * bn: (j + 1) * bn].sum(1).to(torch.float32)
        if d.C_f32:
            pm.view(d.C_f32, M, d.N, d.ldc).copy_(v[:, : d.N].to(torch.float32))
        return
    v = v[:, : 32 * d.c_kbn]
    Cv = planes_view(pm, d.C_planes, npan, d.c_nkb, fmt)
    if d.residual:
        R = planes_decode(planes_view(pm, d.residual, npan, d.c_nkb, fmt), M)[:, 32 * d.c_kb0: 32 * (d.c_kb0 + d.c_kbn)]
        v = R.to(dtype) + d.res_sign * v
    if fmt == 1 and d.range_flag and not bool((v.abs() < 65000.0).all()):
        pm.view(d.range_flag, 1, 1, 1, dtype=torch.int32)[0, 0] = 1
    planes_encode(Cv, v.to(torch.float32), d.c_kb0)


def emulate_coupling_planes(d, pm: PtrMap, dtype=torch.float32):
    M, fmt = d.M, d.format
    npl = 2 if fmt == 1 else 3
    npan = -(-M // 16)
    zv = planes_view(pm, d.z, npan, d.z_nkb, fmt)
    Z = planes_decode(zv, M)

    def weights(ptr, rows, ld, plane, K):
        Wp = pm.view(ptr, npl, rows * ld, plane).view(npl, rows, ld)
        W = Wp[0].float() + Wp[1].float()
        if npl == 3:
            W = W + Wp[2].float()
        slot = torch.tensor([32 * (c // 32) + _SLOT_OF_FEATURE[c % 32] for c in range(K)])
        return W[:, :K][:, slot]

    gated = d.act == _ext.ACT_GATE

    def act(v, l):
        if gated:      # leaky_relu_backward from the saved activations (only plane 0 of gate[l] is read)
            hv = planes_view(pm, d.gate[l], npan, 8, fmt)[:, :, :1]
            h0 = planes_decode(torch.cat([hv, torch.zeros_like(hv)], dim=2), M)
            v = torch.where(h0.to(dtype) > 0, v, v * d.slope)
        elif d.act == _ext.ACT_LEAKY_RELU:
            v = torch.where(v > 0, v, v * d.slope)
        if l < 2 and d.hidden_out[l]:
            planes_encode(planes_view(pm, d.hidden_out[l], npan, 8, fmt), v.to(torch.float32), 0)
        return v

    A = Z[:, 32 * d.kb_p0: 32 * (d.kb_p0 + d.nk_p)].to(dtype)
    h = act(A @ weights(d.W_in, 256, d.ldw_in, d.w_in_plane, 32 * d.nk_p).to(dtype).t() + pm.vec(d.b_in, 256).to(dtype), 0)
    for j in range(d.n_hidden - 1):
        h = act(h @ weights(d.W_hid[j], 256, d.ldw_hid, d.w_hid_plane, 256).to(dtype).t() + pm.vec(d.b_hid[j], 256).to(dtype), j + 1)
    out = h @ weights(d.W_out, 32 * d.nk_t, d.ldw_out, d.w_out_plane, 256).to(dtype).t() + pm.vec(d.b_out, 32 * d.nk_t).to(dtype)
    v = Z[:, 32 * d.kb_t0: 32 * (d.kb_t0 + d.nk_t)].to(dtype) + d.sign * out
    if fmt == 1 and d.range_flag and not bool((v.abs() < 65000.0).all() and (h.abs() < 65000.0).all()):
        pm.view(d.range_flag, 1, 1, 1, dtype=torch.int32)[0, 0] = 1
    planes_encode(zv, v.to(torch.float32), d.kb_t0)


# ---- direct calls of the planes entry points (the training backward on the planes pipeline, training.py) ----------------
def _tensor_planes_view(t, M, nkb):
    npan = -(-M // 16)
    return t.view(-1)[: npan * nkb * 3072].view(torch.bfloat16).view(npan, nkb, 3, 64, 8)


def _emu_pack_planes_call(src, planes, *, M, nkb, idx, ld=None, pre_div=None, pre_sub=None, fmt=0, range_flag=None, src_cols=0,
                          grad=None):
    ld = src.stride(0) if ld is None else ld
    idx = idx.long()
    rows = torch.as_strided(src, (M, int(idx.max()) + 1), (ld, 1), src.storage_offset())
    if grad is not None:          # g = row_weight * d/dz base(z) (usf_base_logprob_grad_f32's Laplace / Normal formulas)
        base, w, loc, scale = grad
        n = rows.shape[1]
        t = rows - loc[:n]
        rows = (-torch.sign(t) / scale[:n] if base == _ext.BASE_LAPLACE else -t / (scale[:n] * scale[:n])) * w[:M, None]
    X = torch.zeros(M, 32 * nkb)
    ok = idx >= 0
    X[:, ok] = rows[:, idx[ok]]
    if pre_div is not None:
        X[:, ok] = X[:, ok] / pre_div[ok]
    if pre_sub is not None:
        X[:, ok] = X[:, ok] - pre_sub[ok]
    planes_encode(_tensor_planes_view(planes, M, nkb), X, 0)


def _planes_weight(W_planes, K):
    W = W_planes[0].float() + W_planes[1].float() + W_planes[2].float()
    slot = torch.tensor([32 * (c // 32) + _SLOT_OF_FEATURE[c % 32] for c in range(K)])
    return W[:, :K][:, slot]


def _emu_gemm_planes_call(A, W_planes, *, M, a_nkb, nk, a_kb0=0, bias=None, post_mul=None, residual=None, C_planes=None, c_nkb=0,
                          c_kb0=0, c_kbn=0, C_f32=None, ldc=0, N=0, res_sign=1.0, act=0, slope=0.0, fmt=0, range_flag=None):
    assert fmt == 0 and residual is None and C_f32 is None and post_mul is None
    Am = planes_decode(_tensor_planes_view(A, M, a_nkb), M)[:, 32 * a_kb0: 32 * (a_kb0 + nk)].double()
    v = Am @ _planes_weight(W_planes, 32 * nk).double().t()
    if bias is not None:
        v = v + bias.double()
    if act == _ext.ACT_LEAKY_RELU:
        v = torch.where(v > 0, v, v * slope)
    planes_encode(_tensor_planes_view(C_planes, M, c_nkb), v[:, : 32 * c_kbn].float(), c_kb0)


def _emu_wgrad_blocked(Yp, y_nkb, y_kb0, Ap, a_nkb, a_kb0, G, *, M, N, K, ldg, g_off=0, alpha=1.0, beta=0.0, colsum=None,
                       cs_alpha=1.0, cs_beta=0.0, queue=None, ws=None):
    Y = planes_decode(_tensor_planes_view(Yp, M, y_nkb), M)[:, 32 * y_kb0: 32 * y_kb0 + N].double()
    A = planes_decode(_tensor_planes_view(Ap, M, a_nkb), M)[:, 32 * a_kb0: 32 * a_kb0 + K].double()
    g = _view(G, g_off, N, K, ldg)
    r = alpha * (Y.t() @ A)
    g.copy_((r + beta * g.double() if beta != 0.0 else r).float())
    if colsum is not None:
        c = cs_alpha * Y.sum(0)
        cv = colsum.view(-1)[:N]
        cv.copy_((c + cs_beta * cv.double() if cs_beta != 0.0 else c).float())


_LAST_RUN = {}


def _emu_coupling_op(op, device):
    """_ext.coupling_op of the training backward (the fused / tiny-layer kernel in gate mode): every tensor the descriptor can
    point into -- workspace, packed weights of the forward and backward images"""
    eng, plan = _LAST_RUN["eng"], _LAST_RUN["plan"]
    pm = PtrMap()
    for t in plan["ws"].values():
        pm.add(t)
    for cp in plan["pk"]["coupling"].values():
        for key in ("fused", "fused_bwd"):
            f = cp.get(key)
            if f:
                for t in [f.get("W_in"), f.get("b_in"), f.get("W_out"), f.get("b_out"), f.get("zeros"), f.get("W_ctx"), f.get("b_ctx")]:
                    pm.add(t)
                for hv in f.get("hid", []):
                    for t in (hv if isinstance(hv, (tuple, list)) else (hv,)):
                        pm.add(t)
    emulate_coupling(op.u.coupling, pm, torch.float64)


def _emu_coupling_planes_op(op, device):
    eng, plan = _LAST_RUN["eng"], _LAST_RUN["plan"]
    pm = PtrMap()
    for t in plan["ws"].values():
        pm.add(t)
    for group in ("mats", "vecs"):
        for t in plan["pk"][group].values():
            pm.add(t)
    for cp in plan["pk"]["coupling"].values():
        if "planes_bwd" in cp:
            pm.add(cp["planes_bwd"]["zeros"])
    emulate_coupling_planes(op.u.coupling_planes, pm, torch.float64)


def _gather(src, dst, idx):
    out = torch.zeros(src.shape[0], idx.numel())
    ok = idx >= 0
    out[:, ok] = src[:, idx[ok].long()]
    dst.copy_(out)


def run_plan(eng, plan, x, out, context=None, dtype=torch.float32):
    """CPU stand-in for FlowEngine._execute."""
    _LAST_RUN.update(eng=eng, plan=plan)
    ws, pk = plan["ws"], plan["pk"]
    pm = PtrMap()
    for t in ws.values():
        pm.add(t)
    for t in eng.__dict__.get("_idx_cache", {}).values():
        pm.add(t)
    for group in ("mats", "vecs"):
        for t in pk[group].values():
            pm.add(t)
    for cp in pk["coupling"].values():
        if "unfused" in cp:
            u = cp["unfused"]
            for W, b in u.get("layers", []):
                pm.add(W), pm.add(b)
            for t in [u["W_out"], u["b_out"], u.get("W_ctx4"), u.get("b_ctx")] + u.get("tensors", []):
                pm.add(t)
        if "fused" in cp:
            f = cp["fused"]
            for t in (f["W_in"], f["b_in"], f["W_out"], f["b_out"], f.get("W_ctx"), f.get("b_ctx")):
                pm.add(t)
            for W, b in f["hid"]:
                pm.add(W), pm.add(b)
    pm.add(x)
    pm.add(out)
    B = x.shape[0]
    if context is not None:
        ws["ctx4"][:, 0].copy_(context.reshape(B))
        ws["ctx"].copy_(context.reshape(B))
    arr = plan["arr"]
    for idx, member, field in plan["patch_in"]:
        setattr(getattr(arr[idx].u, member), field, x.data_ptr())
    for idx, member, field in plan["patch_out"]:
        setattr(getattr(arr[idx].u, member), field, out.data_ptr())
    pos = 0

    def run_until(end):
        nonlocal pos
        while pos < end:
            op = arr[pos]
            if op.kind == _ext.OP_LINEAR:
                emulate_linear(op.u.linear, pm, dtype)
            elif op.kind == _ext.OP_PACK_PLANES:
                emulate_pack_planes(op.u.pack_planes, pm, dtype)
            elif op.kind == _ext.OP_GEMM_PLANES:
                emulate_gemm_planes(op.u.gemm_planes, pm, dtype)
            elif op.kind == _ext.OP_COUPLING_PLANES:
                emulate_coupling_planes(op.u.coupling_planes, pm, dtype)
            elif op.kind == _ext.OP_GATED_NORM:
                emulate_gated_norm(op.u.gated_norm, pm, dtype)
            else:
                emulate_coupling(op.u.coupling, pm, dtype)
            pos += 1

    for g in plan["side"]:
        run_until(g[1])
        if g[0] == "scale":
            _, _, buf, ld, sc, divide, ncols = g
            ws[buf][:, :ncols] = ws[buf][:, :ncols] / sc if divide else ws[buf][:, :ncols] * sc
        else:
            _, _, src, dst_name, dst_layout = g
            src_t = x if src[0] == "user_in" else ws[src[0]]
            _gather(src_t, ws[dst_name], eng._gather_index(src[1], dst_layout, x.device))
    run_until(plan["n"])
    fg = plan["final_gather"]
    if fg is not None:
        src, dst_name = fg
        if dst_name == "user_out":
            _gather(ws[src[0]], out, eng._gather_index(src[1], "user", x.device))
        else:
            _gather(ws[src[0]], ws[dst_name], eng._gather_index(src[1], "nat", x.device))


# ---- parameter-prep entry points (usf_prep.hip), documented semantics of include/usflows_hip.h ----------
def _emu_lu_prepare(L_raws, U_raws, want_M=True, want_Minv=True, keep_factors=False):
    n, D = len(L_raws), L_raws[0].shape[0]
    eye = torch.eye(D, dtype=torch.float64)
    tri = torch.zeros(2 * n, D, D, dtype=torch.float64)
    inv = torch.zeros_like(tri)
    for i, (L, U) in enumerate(zip(L_raws, U_raws)):
        tri[2 * i] = L.double().tril(-1) + eye
        tri[2 * i + 1] = U.double().triu().t()
        inv[2 * i] = torch.linalg.solve_triangular(tri[2 * i], eye, upper=False)
        inv[2 * i + 1] = torch.linalg.solve_triangular(tri[2 * i + 1], eye, upper=False)
    out = dict(M=(tri[0::2] @ tri[1::2].transpose(1, 2)) if want_M else None,
               Minv=(inv[1::2].transpose(1, 2) @ inv[0::2]) if want_Minv else None,
               ladj=torch.stack([U.double().diagonal().abs().log().sum() for U in U_raws]))
    if keep_factors:
        out["tri"], out["tri_inv"] = tri, inv
    return out


def _emu_householder(w_0, vk, out=None):
    w = w_0.double()
    for v in vk.double():
        w = w - 2.0 * torch.outer(w @ v, v) / torch.dot(v, v)
    if out is not None:
        out.copy_(w)
        return out
    return w.contiguous()


def _emu_matmul_f64(A, B, transA=False, transB=False, tri=0):
    return ((A.t() if transA else A) @ (B.t() if transB else B)).contiguous()


def _bf16_planes(w32):
    hi = w32.to(torch.bfloat16)
    r = w32 - hi.float()
    mid = r.to(torch.bfloat16)
    return hi, mid, (r - mid.float()).to(torch.bfloat16)


def _emu_pack_weight(src, out_idx, n_out, in_idx, n_in, *, W=None, ldw=0, planes=None, transpose=False, ld_src=None):
    if _ext._tls.jobs and n_out > 0 and n_in > 0:
        # queued like the real call inside a batch_jobs block: runs at the flush, BEHIND the queued gradient jobs
        _ext._tls.jobs[-1].jobs.append(lambda: _emu_pack_weight_now(src, out_idx, n_out, in_idx, n_in, W=W, ldw=ldw, planes=planes,
                                                                  transpose=transpose, ld_src=ld_src))
        return
    _emu_pack_weight_now(src, out_idx, n_out, in_idx, n_in, W=W, ldw=ldw, planes=planes, transpose=transpose, ld_src=ld_src)


def _emu_flush(self):
    """batch_jobs.flush of the emulation: the queued gradient jobs first, then the queued pack jobs (the real order)"""
    gj, self.grad_jobs = self.grad_jobs, []
    for fn in gj:
        fn()
    jobs, self.jobs = self.jobs, []
    for fn in jobs:
        fn()


def _emu_pack_weight_now(src, out_idx, n_out, in_idx, n_in, *, W=None, ldw=0, planes=None, transpose=False, ld_src=None):
    ld = ld_src if ld_src is not None else src.shape[-1]
    S = torch.as_strided(src.reshape(-1), (src.numel() // ld, ld), (ld, 1))
    if transpose:
        S = S.t()
    oi = torch.arange(n_out) if out_idx is None else out_idx[:n_out].long()
    ii = torch.arange(n_in) if in_idx is None else in_idx[:n_in].long()
    v = torch.zeros(n_out, n_in, dtype=torch.float32)
    ro, ci = torch.nonzero(oi >= 0).flatten(), torch.nonzero(ii >= 0).flatten()
    v[ro[:, None], ci[None, :]] = S[oi[ro][:, None], ii[ci][None, :]].float()
    if W is not None:
        torch.as_strided(W.reshape(-1), (n_out, n_in), (ldw, 1)).copy_(v)
    if planes is not None and planes.dtype == torch.float16:           # fp16x2 planes (transpose bit 1 of the C entry)
        planes.zero_()
        hi = v.to(torch.float16)
        planes[0, :n_out, :n_in] = hi
        planes[1, :n_out, :n_in] = (v - hi.float()).to(torch.float16)
    elif planes is not None:
        planes.zero_()
        for q, pl in enumerate(_bf16_planes(v)):
            planes[q, :n_out, :n_in] = pl


def _emu_matvec_f64(src, b, *, idx=None, n_out=None, alpha=1.0, out32=None, out64=None, ld_src=None):
    n_out = src.shape[0] if n_out is None else n_out
    oi = torch.arange(n_out) if idx is None else idx[:n_out].long()
    r = torch.zeros(n_out, dtype=torch.float64)
    ok = oi >= 0
    r[ok] = alpha * (src[oi[ok]] @ b)
    if out32 is not None:
        out32.copy_(r.float())
    if out64 is not None:
        out64.copy_(r)


# ---- batch-sized entry points as the _ext wrappers present them (tensors + element offsets) ---------------
def _view(t, off, rows, cols, ld):
    return torch.as_strided(t.reshape(-1), (rows, cols), (ld, 1), t.storage_offset() + off)   # as_strided offsets are absolute


def _emu_linear(A, W, C_out, *, M, N, K, lda, ldw, ldc, bias=None, pre_div=None, pre_sub=None, residual=None, ldr=0,
                post_mul=None, res_sign=1.0, act=_ext.ACT_NONE, slope=0.0, a_off=0, c_off=0, r_off=0, addend=None,
                ldadd=0, W_split=None, dtype=torch.float64):
    a = _view(A, a_off, M, K, lda).to(dtype)
    if pre_div is not None:
        a = a / pre_div[:K].to(dtype)
    if pre_sub is not None:
        a = a - pre_sub[:K].to(dtype)
    v = a @ _view(W, 0, N, K, ldw).to(dtype).t()
    if bias is not None:
        v = v + bias[:N].to(dtype)
    if act == _ext.ACT_GATE:
        v = torch.where(_view(addend, 0, M, N, ldadd) > 0, v, v * slope)
    elif addend is not None:
        v = v + _view(addend, 0, M, N, ldadd).to(dtype)
    if act == _ext.ACT_LEAKY_RELU:
        v = torch.where(v > 0, v, v * slope)
    if residual is not None:
        v = _view(residual, r_off, M, N, ldr).to(dtype) + res_sign * v
    if post_mul is not None:
        v = v * post_mul[:N].to(dtype)
    _view(C_out, c_off, M, N, ldc).copy_(v.to(torch.float32))


def _emu_wgrad(Y, A, G, *, M, N, K, ldy, lda, ldg, y_off=0, a_off=0, g_off=0, alpha=1.0, beta=0.0, mode=0, defer=True):
    bj = _ext._defer_grad_job(M) if defer else None
    if bj is not None:         # as the real call: queued, reads its operands at the flush (usf_grad_jobs_f32)
        bj.grad_jobs.append(lambda: _emu_wgrad(Y, A, G, M=M, N=N, K=K, ldy=ldy, lda=lda, ldg=ldg, y_off=y_off, a_off=a_off,
                                               g_off=g_off, alpha=alpha, beta=beta, mode=mode, defer=False))
        return
    y = _view(Y, y_off, M, N, ldy).double()
    a = _view(A, a_off, M, K, lda).double()
    g = _view(G, g_off, N, K, ldg)
    g.copy_((alpha * (y.t() @ a) + (beta * g.double() if beta != 0.0 else 0.0)).float())


def _emu_colsum(Y, out, *, M, N, ldy, y_off=0, alpha=1.0, beta=0.0, defer=True):
    bj = _ext._defer_grad_job(M) if defer else None
    if bj is not None:
        bj.grad_jobs.append(lambda: _emu_colsum(Y, out, M=M, N=N, ldy=ldy, y_off=y_off, alpha=alpha, beta=beta, defer=False))
        return
    o = out.reshape(-1)[:N]
    o.copy_((alpha * _view(Y, y_off, M, N, ldy).double().sum(0) + (beta * o.double() if beta != 0.0 else 0.0)).float())


def _emu_gated_norm_rows(skip, *, M, C_cols, c_pad=None, ld_skip=None, vg=None, ld_vg=0, gate_off=0, gamma=None, beta=None, eps=1e-5,
                         out=None, ld_out=0, out_act=None, ld_act=0, act=_ext.ACT_NONE, slope=0.0, dtype=torch.float64):
    """usf_gated_norm_rows_f32 (include/usflows_hip.h) on raw buffers"""
    Cn, Cp = C_cols, c_pad if c_pad is not None else C_cols
    r = _view(skip, 0, M, Cn, ld_skip if ld_skip is not None else Cn).to(dtype)
    if vg is not None:
        r = r + _view(vg, 0, M, Cn, ld_vg).to(dtype) * torch.sigmoid(_view(vg, gate_off, M, Cn, ld_vg).to(dtype))
    if gamma is not None:
        mean = r.mean(dim=1, keepdim=True)
        var = ((r - mean) ** 2).mean(dim=1, keepdim=True)
        r = (r - mean) / torch.sqrt(var + eps) * gamma[:Cn].to(dtype) + beta[:Cn].to(dtype)
    full = torch.zeros(M, Cp, dtype=dtype)
    full[:, :Cn] = r
    if out is not None:
        _view(out, 0, M, Cp, ld_out).copy_(full.float())
    if out_act is not None:
        a = torch.where(full > 0, full, full * slope) if act == _ext.ACT_LEAKY_RELU else full
        _view(out_act, 0, M, Cp, ld_act).copy_(a.float())


def _emu_gated_norm_rows_bwd(skip, dy, d_skip, *, M, C_cols, c_pad, ld_skip, ld_dy, ld_d_skip, vg=None, ld_vg=0, gate_off=0, d_vg=None,
                             ld_d_vg=0, gamma=None, eps=1e-5, dy_xh=None, ld_dy_xh=0, dtype=torch.float64):
    """usf_gated_norm_rows_bwd_f32 (include/usflows_hip.h): the formulas of the header, not autograd"""
    Cn, Cp = C_cols, c_pad
    r = _view(skip, 0, M, Cn, ld_skip).to(dtype)
    val = sg = None
    if vg is not None:
        val = _view(vg, 0, M, Cn, ld_vg).to(dtype)
        sg = torch.sigmoid(_view(vg, gate_off, M, Cn, ld_vg).to(dtype))
        r = r + val * sg
    g = _view(dy, 0, M, Cn, ld_dy).to(dtype)
    if gamma is not None:
        mean = r.mean(dim=1, keepdim=True)
        rstd = 1.0 / torch.sqrt(((r - mean) ** 2).mean(dim=1, keepdim=True) + eps)
        xh = (r - mean) * rstd
        if dy_xh is not None:
            full = torch.zeros(M, Cp, dtype=dtype)
            full[:, :Cn] = g * xh
            _view(dy_xh, 0, M, Cp, ld_dy_xh).copy_(full.float())
        g = g * gamma[:Cn].to(dtype)
        g = (g - g.mean(dim=1, keepdim=True) - xh * (g * xh).mean(dim=1, keepdim=True)) * rstd
    full = torch.zeros(M, Cp, dtype=dtype)
    full[:, :Cn] = g
    _view(d_skip, 0, M, Cp, ld_d_skip).copy_(full.float())
    if d_vg is not None:
        a, b = torch.zeros(M, Cp, dtype=dtype), torch.zeros(M, Cp, dtype=dtype)
        a[:, :Cn] = g * sg
        b[:, :Cn] = g * val * sg * (1 - sg)
        _view(d_vg, 0, M, Cp, ld_d_vg).copy_(a.float())
        _view(d_vg, gate_off, M, Cp, ld_d_vg).copy_(b.float())


def _emu_add_rows(x, t, ones):
    x.add_(t)


def _emu_act_grad(d, h, *, M, H, ldd, ldh, act, slope):
    if act == _ext.ACT_NONE:
        return
    dv, hv = _view(d, 0, M, H, ldd), _view(h, 0, M, H, ldh)
    dv.copy_(torch.where(hv > 0, dv, dv * slope))


def _emu_base_logprob(z, ldz, M, D, base, loc, scale, logdet_const, out, sum_out=None, logdet_dev=None):
    if logdet_dev is not None:
        logdet_const = float(logdet_const) + float(logdet_dev.float())
    zz = _view(z, 0, M, D, ldz).double()
    if base == _ext.BASE_ROWSUM:
        out.copy_((zz.sum(-1) + logdet_const).float())
        if sum_out is not None:
            sum_out[0] += out.double().sum()
            sum_out[1] += M
        return
    if base in (_ext.BASE_LPNORM1, _ext.BASE_LPNORM2, _ext.BASE_LPNORMINF):
        p = {_ext.BASE_LPNORM1: 1.0, _ext.BASE_LPNORM2: 2.0}.get(base, float("inf"))
        out.copy_((zz - loc.double()).norm(p=p, dim=1).float())
        return
    dist = (torch.distributions.Laplace if base == _ext.BASE_LAPLACE else torch.distributions.Normal)(loc.double(), scale.double())
    out.copy_((dist.log_prob(zz).sum(-1) + logdet_const).float())


def _emu_base_tables(base, loc, scale, D, tab, stride):
    t = tab.view(3, stride)
    t.zero_()
    t[0, :D] = loc
    t[1, :D] = 1.0 / scale
    t[2, :D] = -(2.0 * scale).log() if base == _ext.BASE_LAPLACE else -scale.log() - 0.91893853320467274178


def _emu_base_logprob_grad(z, ldz, g_lp, M, D, base, loc, scale, g, ldg):
    t = _view(z, 0, M, D, ldz).double() - loc.double()
    if base == _ext.BASE_LAPLACE:
        v = -torch.sign(t) / scale.double()
    elif base == _ext.BASE_NORMAL:
        v = -t / (scale.double() ** 2)
    elif base == _ext.BASE_LPNORM1:
        v = torch.sign(t)
    elif base == _ext.BASE_LPNORM2:
        v = t / t.norm(dim=1, keepdim=True)
    else:
        v = torch.sign(t) * (t.abs() == t.abs().max(dim=1, keepdim=True).values)
    gv = _view(g, 0, M, ldg, ldg)
    gv.zero_()
    gv[:, :D] = (v * g_lp.double()[:, None]).float()


def _emu_base_param_grad(z, ldz, g_lp, M, D, base, loc, scale, out):
    zz = _view(z, 0, M, D, ldz).double()
    t = zz - loc.double()
    b = scale.double()
    w = g_lp.double()[:M, None]
    if base == _ext.BASE_LAPLACE:
        dl, ds = torch.sign(t) / b, t.abs() / (b * b) - 1.0 / b
    else:
        dl, ds = t / (b * b), t * t / (b * b * b) - 1.0 / b
    out[0, :D].copy_((w * dl).sum(0).float())
    out[1, :D].copy_((w * ds).sum(0).float())


def _emu_gemm_f64(A, B, Cout, *, M, N, K, lda, ldb, ldc, transA=False, transB=False, batch=1, strideA=0, strideB=0,
                  strideC=0, alpha=1.0, beta=0.0, tri=0, a_off=0, b_off=0, c_off=0):
    for i in range(batch):
        a = _view(A, a_off + i * strideA, K if transA else M, M if transA else K, lda)
        b = _view(B, b_off + i * strideB, N if transB else K, K if transB else N, ldb)
        c = _view(Cout, c_off + i * strideC, M, N, ldc)
        r = alpha * ((a.t() if transA else a) @ (b.t() if transB else b))
        c.copy_(r + beta * c if beta != 0.0 else r)


def _emu_lu_grad_finish(dL, dU, TL, TU, c, tri, n, D, out_L, out_U):
    l = dL[:n].double() + (TL[:n].double() if TL is not None else 0.0)
    u = dU[:n].double().triu()
    u = u + torch.diag_embed(c[:n, None].double() / tri[1:2 * n:2].double().diagonal(dim1=1, dim2=2))
    if TU is not None:
        u = u + TU[:n].double().triu()
    out_L.view(n, D, D).copy_(l.tril(-1).float())
    out_U.view(n, D, D).copy_(u.float())


def install_training_emulation(monkeypatch):
    """parameter prep + every batch-sized entry point of the training path on torch-CPU; FlowEngine runs its op
    lists through run_plan.  Exercises engine.py / training.py (layouts, index maps, chain rule) without a GPU."""
    from usflows_amd.engine import FlowEngine
    install_prep_emulation(monkeypatch)
    monkeypatch.setattr(_ext, "linear", _emu_linear)
    monkeypatch.setattr(_ext, "wgrad", _emu_wgrad)
    monkeypatch.setattr(_ext, "colsum", _emu_colsum)
    monkeypatch.setattr(_ext, "act_grad", _emu_act_grad)
    monkeypatch.setattr(_ext, "gated_norm_rows", _emu_gated_norm_rows)
    monkeypatch.setattr(_ext, "gated_norm_rows_bwd", _emu_gated_norm_rows_bwd)
    monkeypatch.setattr(_ext, "add_rows", _emu_add_rows)
    monkeypatch.setattr(_ext, "base_logprob", _emu_base_logprob)
    monkeypatch.setattr(_ext, "base_tables", _emu_base_tables)
    monkeypatch.setattr(_ext, "base_logprob_grad", _emu_base_logprob_grad)
    monkeypatch.setattr(_ext, "base_param_grad", _emu_base_param_grad)
    monkeypatch.setattr(_ext, "gemm_f64", _emu_gemm_f64)
    monkeypatch.setattr(_ext, "pack_planes", _emu_pack_planes_call)
    monkeypatch.setattr(_ext, "gemm_planes", _emu_gemm_planes_call)
    monkeypatch.setattr(_ext, "wgrad_blocked", _emu_wgrad_blocked)
    monkeypatch.setattr(_ext, "coupling_planes_op", _emu_coupling_planes_op)
    monkeypatch.setattr(_ext, "coupling_op", _emu_coupling_op)
    monkeypatch.setattr(_ext, "lu_grad_finish", _emu_lu_grad_finish)
    monkeypatch.setattr(FlowEngine, "_check_input", lambda self, x: x.contiguous().float())
    monkeypatch.setattr(FlowEngine, "_execute_plain",
                        lambda self, plan, x, out, context: run_plan(self, plan, x, out, context, dtype=torch.float64))
    monkeypatch.setattr(FlowEngine, "_execute",
                        lambda self, plan, x, out, context: run_plan(self, plan, x, out, context, dtype=torch.float64))


def install_prep_emulation(monkeypatch):
    """route the engine's parameter-prep calls to the torch-CPU statements above (tests without a GPU)"""
    monkeypatch.setattr(_ext, "TAPES_ENABLED", False)     # the emulated entry points are not taped: always rebuild
    monkeypatch.setattr(_ext, "lu_prepare", _emu_lu_prepare)
    monkeypatch.setattr(_ext, "householder", _emu_householder)
    monkeypatch.setattr(_ext, "matmul_f64", _emu_matmul_f64)
    monkeypatch.setattr(_ext, "pack_weight", _emu_pack_weight)
    monkeypatch.setattr(_ext.batch_jobs, "flush", _emu_flush)
    monkeypatch.setattr(_ext, "matvec_f64", _emu_matvec_f64)
    monkeypatch.setattr(_ext, "gemm_f64", _emu_gemm_f64)


def engine_transform(eng, x, direction, context=None, fused=False, planes=False):
    eng.use_fused_coupling = fused
    eng.fused_min_rows = 0
    eng.use_planes, eng.planes_min_rows = bool(planes), 0
    if planes:
        eng.gemm_mode = planes if isinstance(planes, str) else "bf16x3"
    if fused:
        eng._fused_ok = lambda cp: len(cp["hidden"]) <= 3
    B = x.shape[0]
    out = torch.empty(B, eng.D)
    plan = eng._plan(direction, B, x.device, context is not None, "user")
    run_plan(eng, plan, x.contiguous(), out, context)
    return out


def engine_latent(eng, x, context=None, fused=False, planes=False):
    eng.use_fused_coupling = fused
    eng.fused_min_rows = 0
    eng.use_planes, eng.planes_min_rows = bool(planes), 0
    if planes:
        eng.gemm_mode = planes if isinstance(planes, str) else "bf16x3"
    if fused:
        eng._fused_ok = lambda cp: len(cp["hidden"]) <= 3
    plan = eng._plan("backward", x.shape[0], x.device, context is not None, "nat")
    run_plan(eng, plan, x.contiguous(), None, context)
    buf = plan["ws"][plan["out_buf"][0]]
    return buf[:, : eng.D].clone(), -float(plan["pk"]["ladj_total"])


def engine_base_log_prob(eng, x, base, loc, scale, fused=False, planes="bf16x3"):
    """Flow.log_prob's planes plan with the base density in the last GEMM's epilogue (Engine.latent_base_sums + the row-sum tail)"""
    eng.use_fused_coupling = fused
    eng.fused_min_rows = 0
    eng.use_planes, eng.planes_min_rows, eng.gemm_mode = True, 0, planes
    if fused:
        eng._fused_ok = lambda cp: len(cp["hidden"]) <= 3
    B = x.shape[0]
    plan = eng._plan("backward", B, x.device, False, f"base{base}")
    ws = plan["ws"]
    _emu_base_tables(base, loc, scale, eng.D, ws["btab"], ws["btab"].numel() // 3)
    run_plan(eng, plan, x.contiguous(), None, None)
    out = torch.empty(B)
    _emu_base_logprob(ws["bpart"], 8, B, plan["n_part"], _ext.BASE_ROWSUM, None, None, -float(plan["pk"]["ladj_total"]), out)
    return out
