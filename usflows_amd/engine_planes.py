"""The planes pipeline's plans: between the dense layers of a flat flow the activations travel as bf16x3
(or fp16x2) planes in a blocked format -- the op lists of ``log_prob`` / ``backward`` / ``_forward`` at large batches and of the
training step on planes (``usf_pack_planes_f32``, ``usf_gemm_planes_bf16x3``, ``usf_coupling_planes``), and the weight / vector images
they read.  A mixin of ``usflows_amd.engine.FlowEngine`` (split out of engine.py in round 5)."""
from __future__ import annotations

from typing import List

import torch

from . import _ext
from .networks import ConditionalDenseNN, ConvNet, DenseNN


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class PlanesPlanMixin:
    """see the module docstring"""

    # ---- planes pipeline (usf_planes.hip) ---------------------------------------------------------
    @staticmethod
    def _slot_feature(s_: int) -> int:
        """feature offset (0..31) held by slot s of a 32-feature block of a planes buffer (include/usflows_hip.h)"""
        return 16 * ((s_ & 7) >> 2) + 4 * (s_ >> 3) + (s_ & 3)

    def _phys(self, logical: torch.Tensor) -> torch.Tensor:
        """reorder a per-logical-position selector (length a multiple of 32) into physical slot order"""
        n = int(logical.numel())
        perm = torch.tensor([32 * (c // 32) + self._slot_feature(c % 32) for c in range(n)], dtype=torch.long)
        return logical[perm]

    def _planes_fmt(self) -> int:
        """activation / weight plane format of the planes pipeline: fp16x2 in "f16x2" mode (three MFMAs per product,
        22 significant bits per operand) unless a pass just overflowed fp16's range, bf16x3 otherwise"""
        return _ext.PLANES_F16X2 if (self.gemm_mode == "f16x2" and not self._f16_overflow) else _ext.PLANES_BF16X3

    def _planes_ok(self, direction: str, B: int, has_ctx: bool, train: bool) -> bool:
        if self.use_planes is None:
            use = self.gemm_mode == "f16x2" or B >= self.planes_min_rows_bf16x3 or self._has_wide_conditioner()
        else:
            use = bool(self.use_planes)
        if train:
            # the training step on the planes pipeline (round 5; training.py `_backward_planes`): log_prob plans in the bf16x3
            # format whose couplings all run as ONE fused launch (conditioners of <= 2 hidden layers up to 256 wide), from
            # train_planes_min_rows rows; every planes buffer below 2 GiB (usf_wgrad_blocked_f32's offsets)
            use = (self.use_train_planes and direction == "backward" and self.gemm_mode == "bf16x3" and self.use_fused_coupling
                   and B >= max(self.train_planes_min_rows, self.fused_min_rows)
                   and (-(-B // 16)) * (self.LDp // 32) * 3072 < 2 ** 31 and self._train_planes_conditioners_ok())
        if (has_ctx or not use or self._general_cond or self.gemm_mode not in ("bf16x3", "f16x2")
                or B < self.planes_min_rows or (-(-B // 16)) * (self.LDp // 32) * 3072 >= 2 ** 32):
            return False
        # (the structure check does not depend on which runs are merged: a merged run is an affine step like its parts)
        prims = self._primitive_ops(direction, merge=False)
        kinds = [p_[0] for p_ in prims]
        for k_, kind in enumerate(kinds):
            if kind == "scale_div" and not (k_ == 0 and len(kinds) > 1 and kinds[1] in ("affine_bwd", "affine_fwd")):
                return False
            if kind == "scale_mul" and not (k_ == len(kinds) - 1 and k_ > 0 and kinds[k_ - 1] == "affine_fwd"):
                return False
        body = [k_ for k_ in kinds if not k_.startswith("scale")]
        # the last layer must be an affine (it writes the fp32 result) and the chain needs at least two GEMM-sized ops
        return len(body) >= 2 and body[-1].startswith("affine")

    def _train_planes_conditioners_ok(self) -> bool:
        # (the weight-gradient kernel carries the bias sums along only for operands of >= 64 columns -- usf_wgrad_planes_colsum_ok --:
        # flows whose halves are narrower keep the fp32-row training path, which is made for them)
        if min(self.n0a, self.n1a) < 64:
            return False
        for s_ in self.steps:
            if s_.kind == "coupling":
                cond = s_.module.conditioner
                if not isinstance(cond, (ConditionalDenseNN, DenseNN)):
                    return False
                widths = [int(w) for w in cond.hidden_dims]
                if len(widths) > 2 or max(widths) > 256:
                    return False
        return True

    def _has_wide_conditioner(self) -> bool:
        """a conditioner wider than 256 or deeper than 3 hidden layers: no fused coupling kernel serves it"""
        for s_ in self.steps:
            if s_.kind == "coupling":
                cond = s_.module.conditioner
                widths = cond.c_hidden if isinstance(cond, ConvNet) else cond.hidden_dims
                if len(widths) > 3 or max(int(w) for w in widths) > 256:
                    return True
        return False

    def _planes_image(self, pk, key, src, out_sel: torch.Tensor, in_sel: torch.Tensor, fmt: int = 0, transpose: bool = False):
        """cached weight planes of src[out_sel][:, in_sel] (-1: zero; transpose: of src[in_sel][:, out_sel]^T), one queued
        launch: [3, rows, cols] bf16 (bf16x3) or [2, rows, cols] fp16 (fp16x2)"""
        mats = pk["mats"]
        key = key + (fmt,) + (("T",) if transpose else ())
        if key not in mats:
            dev = src.device
            n_out, n_in = int(out_sel.numel()), int(in_sel.numel())
            if fmt == _ext.PLANES_F16X2:
                P = torch.empty(2, n_out, n_in, dtype=torch.float16, device=dev)
            else:
                P = torch.empty(3, n_out, n_in, dtype=torch.bfloat16, device=dev)
            _ext.pack_weight(src, out_sel.to(device=dev, dtype=torch.int32), n_out,
                             in_sel.to(device=dev, dtype=torch.int32), n_in, planes=P, transpose=transpose)
            mats[key] = P
        return mats[key]

    # ---- training on the planes pipeline: the backward launches' weight images (training.py `_backward_body_planes`) ----
    def planes_dgrad_image(self, pk, m) -> torch.Tensor:
        """weight planes of an affine layer's data gradient g_in = g_out W as a usf_gemm_planes_bf16x3 operand: rows = the
        positions of the layer's INPUT layout (segp), K axis = the slots of its OUTPUT layout"""
        blk = m["blk"]
        which = "Minv" if m["prim"] == "affine_bwd" else "M"
        out_phys = self._phys(self.natp_idx if m["out_layout"] == "natp" else self.segp_idx)
        return self._planes_image(pk, ("pl_aff_t", id(blk), which, m["out_layout"]), self._affine_entry(pk, blk)[which],
                                  self.segp_idx, out_phys, _ext.PLANES_BF16X3, transpose=True)

    def planes_coupling_bwd(self, pk, m) -> dict:
        """the conditioner's weights for usf_coupling_planes run BACKWARDS (USF_ACT_GATE): W_in = W_last^T [256, slots of the
        transformed blocks], hidden matrices reversed and transposed, W_out = W_first^T [positions of the conditioning blocks,
        256], zero biases"""
        i = m["step"]
        cp = pk["coupling"][i]
        if "planes_bwd" in cp:
            return cp["planes_bwd"]
        raw = cp["raw"]
        h = list(raw["h"])
        layers = [raw["first"]] + list(raw["hidden"]) + [raw["last"]]
        fmt = _ext.PLANES_BF16X3

        def pad256(n_valid):
            t = torch.full((256,), -1, dtype=torch.long)
            t[:n_valid] = torch.arange(n_valid)
            return t
        ft = m["feat_t"][32 * m["kb_t0"]: 32 * (m["kb_t0"] + m["nk_t"])]
        fp = m["feat_p"][32 * m["kb_p0"]: 32 * (m["kb_p0"] + m["nk_p"])]
        dev = raw["device"]
        f = dict(zeros=torch.zeros(max(256, int(fp.numel())), dtype=torch.float32, device=dev), hid=[])
        f["W_in"] = self._planes_image(pk, ("pl_cin_t", i), layers[-1][0], pad256(h[-1]), self._phys(ft), fmt, transpose=True)
        for j in range(len(h) - 1, 0, -1):        # forward hidden matrix j maps layer j - 1 -> j: backwards j -> j - 1
            f["hid"].append(self._planes_image(pk, ("pl_chid_t", i, j), layers[j][0], pad256(h[j - 1]), self._phys(pad256(h[j])),
                                               fmt, transpose=True))
        f["W_out"] = self._planes_image(pk, ("pl_cout_t", i), layers[0][0], fp, self._phys(pad256(h[0])), fmt, transpose=True)
        cp["planes_bwd"] = f
        return f

    def planes_coupling_bwd_op(self, pk, m, g, g_nkb: int, B: int, gates, d_out) -> _ext.Op:
        """ONE launch for the data-gradient chain of a coupling layer's conditioner on the gradient planes buffer g:
        g[:, conditioning blocks] += sign * MLP^T(g[:, transformed blocks]); gates / d_out: planes buffers (8 blocks per panel)
        in the FORWARD's layer order -- the saved activations resp. the gradients at the pre-activations"""
        cp = pk["coupling"][m["step"]]
        f = self.planes_coupling_bwd(pk, m)
        nl = len(cp["hidden"])
        op = _ext.Op()
        op.kind = _ext.OP_COUPLING_PLANES
        c = op.u.coupling_planes
        c.z, c.z_nkb, c.M = g.data_ptr(), g_nkb, B
        c.kb_p0, c.nk_p, c.kb_t0, c.nk_t = m["kb_t0"], m["nk_t"], m["kb_p0"], m["nk_p"]      # the roles of the block ranges swap
        c.n_hidden, c.hidden_padded = nl, 256
        z = f["zeros"].data_ptr()
        Wi = f["W_in"]
        c.W_in, c.ldw_in, c.w_in_plane, c.b_in = Wi.data_ptr(), Wi.shape[2], Wi.shape[1] * Wi.shape[2], z
        for j, Wh in enumerate(f["hid"]):
            c.W_hid[j], c.b_hid[j] = Wh.data_ptr(), z
            c.ldw_hid, c.w_hid_plane = Wh.shape[2], Wh.shape[1] * Wh.shape[2]
        Wo = f["W_out"]
        c.W_out, c.ldw_out, c.w_out_plane, c.b_out = Wo.data_ptr(), Wo.shape[2], Wo.shape[1] * Wo.shape[2], z
        c.sign, c.slope, c.act, c.format, c.range_flag = m["sign"], cp["slope"], _ext.ACT_GATE, _ext.PLANES_BF16X3, 0
        for l in range(nl):
            c.gate[l] = gates[nl - 1 - l].data_ptr()
            c.hidden_out[l] = d_out[nl - 1 - l].data_ptr()
        return op

    def _planes_vec(self, pk, key, src, sel: torch.Tensor, pad: float = 0.0) -> torch.Tensor:
        """cached fp32 vector src[sel] (-1: pad) of length len(sel)"""
        vecs = pk["vecs"]
        if key not in vecs:
            dev = src.device
            n = int(sel.numel())
            if pad == 0.0:
                out = torch.empty(n, dtype=torch.float32, device=dev)
                _ext.pack_weight(src.reshape(1, -1), None, 1, sel.to(device=dev, dtype=torch.int32), n, W=out, ldw=n,
                                 ld_src=src.numel())
                vecs[key] = out
            else:
                sel_dev = sel.to(dev)                    # (once: a host index in the refresh would synchronise every step)
                from .engine import _refreshed                    # (engine imports this module: late import)
                vecs[key] = _refreshed((n,), torch.float32, dev,
                                       lambda o, src=src, sel=sel_dev: o.copy_(self._perm_vec(src.double(), sel, pad)))
        return vecs[key]

    def _build_plan_planes(self, direction: str, B: int, device, final: str, train: bool = False) -> dict:
        """Launch list of the planes pipeline: pack -> (GEMM on planes)* -> GEMM with fp32 output.

        Every layer is the same kernel: an affine block one GEMM, an additive coupling the chain of its conditioner's
        dense layers (the hidden activations make a round trip through HBM / the Infinity Cache as planes; the last
        one adds / subtracts into the transformed half of z IN PLACE, reading the residual from the planes).  The
        buffer z keeps the engine's segment layout [mask==0 | mask==1] padded to whole 32-feature blocks; a coupling
        reads the blocks that hold its conditioning features (zero weights on the others) and rewrites the blocks
        that hold its transformed features (zero rows elsewhere: those values are rewritten unchanged)."""
        pk = self.pack(device)
        ws = self._workspace(B, device)
        prims = self._primitive_ops(direction, merge=not train)     # (the training backward needs every block's own launch)
        npan = -(-B // 16)
        nkb = self.LDp // 32
        segp, natp = self.segp_idx, self.natp_idx
        seg_phys = self._phys(segp)
        Hp = _round_up(self.hmax, 32)
        fmt = self._planes_fmt()
        chunk = 2048 if fmt == _ext.PLANES_F16X2 else 3072        # bytes per (panel, block): NPL planes of 1 KiB
        if "pflag" not in ws:
            ws["pflag"] = torch.zeros(1, dtype=torch.int32, device=device)
        flag = ws["pflag"].data_ptr() if fmt == _ext.PLANES_F16X2 else 0

        def planes_buf(name, blocks):
            if name not in ws or ws[name].numel() < npan * blocks * 3072:
                ws[name] = torch.empty(npan * blocks * 3072, dtype=torch.uint8, device=device)     # (sized for either format)
            return ws[name]

        # training: every affine output keeps a planes buffer of its own (the saved activations of the backward pass,
        # already in operand form: (K + 1) x B x LDp x 6 bytes -- cfg2 at 65536 rows: 10.4 GB of the 288 GB), couplings update
        # theirs in place (their conditioning half -- all the backward needs of them -- is untouched)
        zbufs = [planes_buf("pzA", nkb), planes_buf("pzB", nkb)] if not train else [planes_buf("pz0", nkb), None]
        znames = ["pzA", "pzB"] if not train else ["pz0", None]
        n_z = [1]
        cur = 0
        ops: List[_ext.Op] = []
        patch_in, patch_out = [], []
        meta: List[dict] = []

        def gemm_op(**kw) -> _ext.Op:
            op = _ext.Op()
            op.kind = _ext.OP_GEMM_PLANES
            g = op.u.gemm_planes
            g.M, g.res_sign, g.slope, g.act = B, 1.0, 0.0, _ext.ACT_NONE
            g.format, g.range_flag = fmt, flag
            for k_, v_ in kw.items():
                setattr(g, k_, v_)
            return op

        # ---- head: the caller's fp32 rows -> planes in segment layout (+ x / s - b of the first layer) -------------
        n = len(prims)
        k = 0
        pack = _ext.Op()
        pack.kind = _ext.OP_PACK_PLANES
        d = pack.u.pack_planes
        d.src, d.ld, d.M, d.nkb = 0, self.D, B, nkb
        d.src_cols = self.D                    # (every index of the layout is a feature number: rows are read whole, coalesced)
        d.format, d.range_flag = fmt, flag
        d.idx = self._idx_dev("segp", device).data_ptr()
        d.planes = zbufs[cur].data_ptr()
        first_bias_in_prologue = False
        if prims[0][0] == "scale_div":
            s0 = self._step(prims[0][1])
            d.pre_div = self._planes_vec(pk, ("pl_scale", id(s0.module), "segp"), pk["scale"][id(s0.module)], segp, 1.0).data_ptr()
            if prims[1][0] == "affine_bwd":      # (x / s - b) Minv^T: the bias goes into the head as well
                blk = self._step(prims[1][1]).module
                d.pre_sub = self._planes_vec(pk, ("pl_b", id(blk), "segp"), pk["affine"][id(blk)]["b"], segp).data_ptr()
                first_bias_in_prologue = True
            k = 1
        patch_in.append((len(ops), "pack_planes", "src"))
        ops.append(pack)
        head_scale = self._step(prims[0][1]).module if prims[0][0] == "scale_div" else None
        head_bias_folded = first_bias_in_prologue

        while k < n:
            prim, i = prims[k]
            s = self._step(i)
            nxt = prims[k + 1] if k + 1 < n else None
            if prim in ("affine_fwd", "affine_bwd"):
                blk = s.module
                a = self._affine_entry(pk, blk)
                fuse_post = prim == "affine_fwd" and nxt is not None and nxt[0] == "scale_mul"
                is_last = (k == n - 1) or (fuse_post and k == n - 2)
                if train:
                    is_head = not any(m_["kind"] == "affine" for m_ in meta) and not any(m_["kind"] == "coupling" for m_ in meta)
                    meta.append(dict(kind="affine", op=len(ops), prim=prim, blk=blk, in_buf=znames[cur], in_layout="segp",
                                     out_layout="natp" if is_last else "segp", N=self.D if is_last else self.LD, K=self.LD,
                                     pre_scale=head_scale if is_head else None, post_scale=None,
                                     pre_sub_folded=bool(is_head and head_bias_folded), is_last=is_last))
                out_sel = natp if is_last else segp
                which = "Minv" if prim == "affine_bwd" else "M"
                W = self._planes_image(pk, ("pl_aff", id(blk), which, is_last), a[which], out_sel, seg_phys, fmt)
                kw = dict(A=zbufs[cur].data_ptr(), a_nkb=nkb, a_kb0=0, nk=nkb, W_planes=W.data_ptr(), ldw=W.shape[2],
                          w_plane_stride=W.shape[1] * W.shape[2], w_rows=W.shape[1])
                lay = "natp" if is_last else "segp"
                if prim == "affine_bwd":
                    if first_bias_in_prologue:
                        first_bias_in_prologue = False        # (x / s - b) @ Minv^T: bias already subtracted by the head
                    else:
                        # (y - b) @ Minv^T == y @ Minv^T + c, c = -(Minv b) formed in fp64 at pack time ("bias folding")
                        if "c" not in a:
                            a["c"] = torch.empty(a["b"].shape, dtype=torch.float64, device=a["b"].device)
                            # (launched now, in front of the queued image jobs: it reads the prepared M^-1 and b only, and the job that
                            # packs c runs with the batch behind the layout loop -- no flush: the images of ALL layers stay one batch)
                            _ext.matvec_f64(a["Minv"], a["b"].contiguous(), alpha=-1.0, out64=a["c"])
                        kw["bias"] = self._planes_vec(pk, ("pl_c", id(blk), lay), a["c"], out_sel).data_ptr()
                else:
                    kw["bias"] = self._planes_vec(pk, ("pl_b", id(blk), lay), a["b"], out_sel).data_ptr()
                if is_last:
                    if fuse_post:
                        s2 = self._step(nxt[1])
                        kw["post_mul"] = self._planes_vec(pk, ("pl_scale", id(s2.module), "natp"),
                                                          pk["scale"][id(s2.module)], natp, 1.0).data_ptr()
                        k += 1
                    if final == "user":
                        kw.update(C_f32=0, ldc=self.D, N=self.D)
                        patch_out.append((len(ops), "gemm_planes", "C_f32"))
                        out_buf = ("user_out", "nat", self.D)
                    elif final.startswith("base"):
                        # Flow.log_prob: z only feeds the base density -- the epilogue reduces the row's Laplace / Normal terms per
                        # column block ([B, 8] partial sums; tables refreshed per call by Engine.latent) and stores no rows
                        stride = _round_up(self.D, 4)
                        if "btab" not in ws:
                            ws["btab"] = torch.zeros(3 * stride, dtype=torch.float32, device=device)
                            ws["bpart"] = torch.zeros(B, 8, dtype=torch.float32, device=device)
                        kw.update(C_f32=0, ldc=self.D, N=self.D, base_tab=ws["btab"].data_ptr(), base_tab_stride=stride,
                                  base_part=ws["bpart"].data_ptr(), base=int(final[4:]))
                        out_buf = ("bpart", "part", 8)
                    else:
                        if "nat2" not in ws:
                            ws["nat2"] = torch.zeros(B, self.LDn, dtype=torch.float32, device=device)
                        kw.update(C_f32=ws["nat2"].data_ptr(), ldc=self.LDn, N=self.D)
                        out_buf = ("nat2", "nat", self.LDn)
                else:
                    if train:
                        znames[1 - cur] = f"pz{n_z[0]}"
                        zbufs[1 - cur] = planes_buf(znames[1 - cur], nkb)
                        n_z[0] += 1
                    kw.update(C_planes=zbufs[1 - cur].data_ptr(), c_nkb=nkb, c_kb0=0, c_kbn=nkb)
                    cur = 1 - cur
                if train:
                    meta[-1]["out_buf"] = out_buf[0] if is_last else znames[cur]
                ops.append(gemm_op(**kw))
                k += 1
                continue
            # ---- additive coupling: its conditioner MLP as a chain of GEMMs, in place on the transformed blocks ----
            cp = pk["coupling"][i]
            raw = cp["raw"]
            sign = 1.0 if prim == "coupling_fwd" else -1.0
            z = zbufs[cur]
            lo_p, hi_p = cp["pass_off"], cp["pass_off"] + int((raw["pass_idx"] >= 0).sum())
            lo_t, hi_t = cp["tr_off"], cp["tr_off"] + cp["tr_n"]
            kb_p0, kb_p1 = lo_p // 32, -(-hi_p // 32)
            kb_t0, kb_t1 = lo_t // 32, -(-hi_t // 32)
            pos = torch.arange(self.LDp)
            feat_p = torch.where((pos >= lo_p) & (pos < hi_p), segp, torch.full_like(segp, -1))
            feat_t = torch.where((pos >= lo_t) & (pos < hi_t), segp, torch.full_like(segp, -1))
            layers = [raw["first"]] + list(raw["hidden"]) + [raw["last"]]
            h = list(raw["h"])
            if (self.use_fused_coupling and B >= self.fused_min_rows and len(h) <= 3 and max(h) <= 256
                    and not (fmt == _ext.PLANES_BF16X3 and len(h) == 3)):
                # ONE launch per layer: usf_coupling_planes (hidden activations stay in registers; widths padded to 256)
                def pad256(n_valid):
                    t = torch.full((256,), -1, dtype=torch.long)
                    t[:n_valid] = torch.arange(n_valid)
                    return t
                op = _ext.Op()
                op.kind = _ext.OP_COUPLING_PLANES
                c = op.u.coupling_planes
                c.z, c.z_nkb, c.M = z.data_ptr(), nkb, B
                c.kb_p0, c.nk_p, c.kb_t0, c.nk_t = kb_p0, kb_p1 - kb_p0, kb_t0, kb_t1 - kb_t0
                c.n_hidden, c.hidden_padded = len(h), 256
                Wi = self._planes_image(pk, ("pl_cin", i), layers[0][0], pad256(h[0]), self._phys(feat_p[32 * kb_p0: 32 * kb_p1]), fmt)
                c.W_in, c.ldw_in, c.w_in_plane = Wi.data_ptr(), Wi.shape[2], Wi.shape[1] * Wi.shape[2]
                c.b_in = self._planes_vec(pk, ("pl_cinb", i), layers[0][1], pad256(h[0])).data_ptr()
                for j in range(1, len(h)):
                    Wh = self._planes_image(pk, ("pl_chid", i, j), layers[j][0], pad256(h[j]), self._phys(pad256(h[j - 1])), fmt)
                    c.W_hid[j - 1] = Wh.data_ptr()
                    c.b_hid[j - 1] = self._planes_vec(pk, ("pl_chidb", i, j), layers[j][1], pad256(h[j])).data_ptr()
                    c.ldw_hid, c.w_hid_plane = Wh.shape[2], Wh.shape[1] * Wh.shape[2]
                out_sel = feat_t[32 * kb_t0: 32 * kb_t1]
                Wo = self._planes_image(pk, ("pl_cout", i), layers[-1][0], out_sel, self._phys(pad256(h[-1])), fmt)
                c.W_out, c.ldw_out, c.w_out_plane = Wo.data_ptr(), Wo.shape[2], Wo.shape[1] * Wo.shape[2]
                c.b_out = self._planes_vec(pk, ("pl_coutb", i), layers[-1][1], out_sel).data_ptr()
                c.sign, c.slope, c.act, c.format, c.range_flag = sign, cp["slope"], cp["act"], fmt, flag
                if train:
                    # the lane-local splits of the hidden activations also go to planes buffers of the layer's own (8 blocks:
                    # 2 x B x 256 x 6 bytes per coupling): operands of the conditioner's weight gradients, gates of its backward
                    hnames = [f"pHs{j}_{i}" for j in range(len(h))]
                    for j, hn in enumerate(hnames):
                        c.hidden_out[j] = planes_buf(hn, 8).data_ptr()
                    meta.append(dict(kind="coupling", op=len(ops), step=i, buf=znames[cur], sign=sign, use_ctx=False,
                                     kb_p0=kb_p0, nk_p=kb_p1 - kb_p0, kb_t0=kb_t0, nk_t=kb_t1 - kb_t0, hidden_planes=hnames,
                                     feat_p=feat_p, feat_t=feat_t))
                ops.append(op)
                k += 1
                continue
            if train:
                from .engine import EngineUnsupported
                raise EngineUnsupported("training on the planes pipeline needs the fused coupling launch")
            hbufs = [planes_buf("pH1", Hp // 32), planes_buf("pH2", Hp // 32)]
            src_buf, src_nkb, src_kb0, src_nk = z, nkb, kb_p0, kb_p1 - kb_p0
            in_sel = self._phys(feat_p[32 * kb_p0: 32 * kb_p1])
            for j, (W_, b_) in enumerate(layers):
                last = j == len(layers) - 1
                if last:
                    out_sel = feat_t[32 * kb_t0: 32 * kb_t1]
                else:
                    hj = _round_up(h[j], 32)
                    out_sel = torch.full((hj,), -1, dtype=torch.long)
                    out_sel[: h[j]] = torch.arange(h[j])
                Wimg = self._planes_image(pk, ("pl_mlp", i, j), W_, out_sel, in_sel, fmt)
                bvec = self._planes_vec(pk, ("pl_mlpb", i, j), b_, out_sel)
                kw = dict(A=src_buf.data_ptr(), a_nkb=src_nkb, a_kb0=src_kb0, nk=src_nk, W_planes=Wimg.data_ptr(),
                          ldw=Wimg.shape[2], w_plane_stride=Wimg.shape[1] * Wimg.shape[2], w_rows=Wimg.shape[1],
                          bias=bvec.data_ptr())
                if last:
                    kw.update(C_planes=z.data_ptr(), c_nkb=nkb, c_kb0=kb_t0, c_kbn=kb_t1 - kb_t0, residual=z.data_ptr(),
                              res_sign=sign)
                else:
                    dst = hbufs[j % 2]
                    kw.update(C_planes=dst.data_ptr(), c_nkb=Hp // 32, c_kb0=0, c_kbn=hj // 32, act=cp["act"],
                              slope=cp["slope"])
                    src_buf, src_nkb, src_kb0, src_nk = dst, Hp // 32, 0, hj // 32
                    hsel = torch.full((hj,), -1, dtype=torch.long)
                    hsel[: h[j]] = torch.arange(h[j])
                    in_sel = self._phys(hsel)
                ops.append(gemm_op(**kw))
            k += 1

        arr = (_ext.Op * len(ops))(*ops)
        n_part = 0
        if out_buf[1] == "part":
            tn = (_ext.load().usf_gemm_planes_variant(arr[len(ops) - 1].u.gemm_planes) - 5000) // 10
            n_part = -(-self.D // (32 * tn))
        return dict(arr=arr, n=len(ops), patch_in=patch_in, patch_out=patch_out, side=[], final_gather=None, n_part=n_part,
                    out_buf=out_buf, ws=ws, pk=pk, meta=meta, planes=True, planes_fmt=fmt, planes_train=bool(train))
