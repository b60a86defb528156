"""Device backward of image-shaped flows (SURVEY rows N2 x N4): what torch autograd computes when ``Flow.fit``
(flows.py:113-210) trains a flow whose layers are the 1 x 1-convolution ``BlockAffineTransform`` (transforms.py:904-962)
and ``MaskedCoupling`` over a ``ConvNet2D`` conditioner (networks.py:40-122, 405-510).

Each ``torch.autograd.Function`` here is one device pass of the inference path with its gradient on the HIP kernels:

* data gradients reuse the forward kernels on the transposed (and, for 3 x 3, flipped) weight:
  ``usf_conv2d_same_f32``, ``usf_pointwise_conv_f32``, ``usf_channel_affine_f32``;
* weight / bias gradients come from ``usf_conv_wgrad_f32`` (exact fp32 on the f32 matrix instruction, deterministic);
* ``usf_layernorm_channels_bwd_f32``, ``usf_gated_residual_bwd_f32``, ``usf_act_grad_f32``, ``usf_masked_residual_f32`` and
  ``usf_base_logprob_grad_f32`` cover the elementwise pieces.

autograd only orders the calls; the tiny parameter maps (LU / Householder factors -> the C x C block matrix and its
inverse) stay torch ops on C x C tensors, as in the reference.
"""
from __future__ import annotations

import threading

import torch

from . import _ext
from .config import config


def _act(a):
    return a if a is not None else (_ext.ACT_NONE, 0.0)


def _gate_inplace(d: torch.Tensor, h: torch.Tensor, act) -> None:
    """d *= act'(h) for a (Leaky)ReLU (sign of the pre-activation == sign of the output)"""
    if act is None or act[0] == _ext.ACT_NONE:
        return
    B = d.shape[0]
    n = d.numel() // B
    _ext.act_grad(d, h, M=B, H=n, ldd=n, ldh=n, act=act[0], slope=act[1])


def conv_shape_ok(conv, B: int, H: int, W: int) -> bool:
    """forward, data gradient (the transposed shape) and weight gradient of this nn.Conv2d at [B, cin, H, W] are all served"""
    if not isinstance(conv, torch.nn.Conv2d):
        return False
    k = conv.kernel_size
    if k[0] != k[1] or k[0] not in (1, 3) or conv.stride != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1 \
            or conv.padding_mode != "zeros":
        return False
    pad = conv.padding
    if not (pad == "same" or (not isinstance(pad, str) and tuple(pad) == (k[0] // 2, k[0] // 2))):
        return False
    lib = _ext.load()
    cin, cout = conv.in_channels, conv.out_channels
    return (lib.usf_conv2d_same_fits(cin, cout, H, W, k[0]) >= 2 and lib.usf_conv2d_same_fits(cout, cin, H, W, k[0]) >= 2
            and lib.usf_conv_wgrad_workspace(max(B, 1), cin, cout, H, W, k[0]) > 0)


def _weight_planes(w, want_transposed: bool, param=None):
    """(planes, planes_t | None) of a convolution weight: both from one launch when the data gradient will be asked for (the
    weight does not change between a step's forward and backward pass); inside ``batched_weight_planes`` from the pass's ONE
    launch for all weights"""
    batch = _WSTATE.planes
    if batch is not None and param is not None:
        hit = batch.get(id(param))
        if hit is not None:
            return hit
    if want_transposed and w.is_cuda and w.dtype == torch.float32:
        return _ext.conv2d_weight_planes_pair(w)
    return _ext.conv2d_weight_planes(w), None


class _WPlanesState(threading.local):
    def __init__(self):
        self.planes = None     # id(weight Parameter) -> (planes, planes_t) of the current pass


_WSTATE = _WPlanesState()


class batched_weight_planes:
    """context manager around one training pass of an image-shaped flow at a launch-bound batch: the bf16x3 plane pairs of ALL
    3 x 3 convolution weights under ``layers`` from one launch (``_ext.WeightPlanesBatch``; the job table is kept on ``owner``
    while the weights keep their addresses) instead of one per convolution and pass"""

    def __init__(self, owner, layers, device):
        self.owner, self.layers, self.device, self.prev = owner, layers, device, None

    def __enter__(self):
        self.prev = _WSTATE.planes
        ws = []
        for l in self.layers:
            for m in l.modules():
                if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and m.weight.device == self.device \
                        and m.weight.dtype == torch.float32 and m.weight.is_contiguous():
                    ws.append(m.weight)
        planes = None
        if ws:
            key = tuple((w.data_ptr(), tuple(w.shape)) for w in ws)
            batch = self.owner.__dict__.get("_wplanes_batch")
            if batch is None or batch.key != key:
                batch = _ext.WeightPlanesBatch([w.detach() for w in ws])
                if not torch.cuda.is_current_stream_capturing():
                    self.owner.__dict__["_wplanes_batch"] = batch
            out = batch.run()
            if out is not None:
                planes = {id(w): pr for w, pr in zip(ws, out)}
        _WSTATE.planes = planes
        return self

    def __exit__(self, *exc):
        _WSTATE.planes = self.prev
        return False


# (Round 4 also tried the weight gradients of a small-batch backward pass on a second stream / a parallel hipGraph branch:
# live MNIST configuration at batch 32 6.25 -> 7.86 ms per replayed step -- every fork / join costs more than the ~5 us launch it
# takes off the chain; profiles/r04_tuning_experiments.md section 3.  The switch and its code are gone.)


def _takeable(params) -> bool:
    """True when the autograd engine will TAKE the gradients of these Parameters as they are handed over (no ``.grad`` yet, no
    hooks that would read them): a gradient whose last sum is still queued (_ext.conv_wgrad(defer=True)) must not be read
    before the backward pass ends"""
    return all(q is None or (q.grad is None and q.is_leaf and not q._backward_hooks
                             and not getattr(q, "_post_accumulate_grad_hooks", None)) for q in params)


class ConvSame(torch.autograd.Function):
    """out_act(bias + conv(in_act(x) * in_mul)) on usf_conv2d_same_f32 (kernel 1 or 3, stride 1, "same")"""

    @staticmethod
    def forward(ctx, x, weight, bias, in_mul, in_act, out_act):
        x = x.contiguous()
        w = weight.detach()
        ks = w.shape[2]
        ia, oa = _act(in_act), _act(out_act)
        planes, ctx.planes_t = _weight_planes(w, ctx.needs_input_grad[0], weight)
        y = _ext.conv2d_same(x, planes, w.shape[0], ks, bias=None if bias is None else bias.detach().contiguous(),
                             in_mul=in_mul, in_act=ia[0], in_slope=ia[1], out_act=oa[0], out_slope=oa[1])
        ctx.save_for_backward(x, w, in_mul, y if out_act is not None else None)
        ctx.cfg = (ks, in_act, out_act, bias is not None)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, in_mul, y = ctx.saved_tensors
        ks, in_act, out_act, has_bias = ctx.cfg
        dy = dy.contiguous()
        if out_act is not None:
            dy = dy.clone()
            _gate_inplace(dy, y, out_act)
        ia = _act(in_act)
        dW = db = dx = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            r = _ext.conv_wgrad(x, dy, ks, in_mul=in_mul, in_act=ia[0], in_slope=ia[1], want_bias=has_bias,
                                defer=_takeable(ctx.params), owners=tuple(id(q) for q in ctx.params if q is not None))
            if r is None:
                raise RuntimeError("usflows_amd: usf_conv_wgrad_f32 does not serve this shape (conv_shape_ok was not consulted)")
            dW, db = r
        if ctx.needs_input_grad[0]:
            planes_t = ctx.planes_t if ctx.planes_t is not None else _ext.conv2d_weight_planes(w, transposed=True)
            dx = None
            if in_act is not None or in_mul is not None:
                # the input's (Leaky)ReLU and mask factors in the data-gradient convolution's output stream
                dx = _ext.conv2d_same_gate(dy, planes_t, w.shape[1], ks, x, ia[1] if in_act is not None else 1.0, in_mul)
            if dx is None:
                dx = _ext.conv2d_same(dy, planes_t, w.shape[1], ks)
                _gate_inplace(dx, x, in_act)
                if in_mul is not None:
                    dx = _ext.masked_residual(None, dx, in_mul, 1.0)
        return dx, dW, db, None, None, None


class ConvSameRes(torch.autograd.Function):
    """rx + sign * om * (conv(x) + bias): a conditioner's LAST convolution with MaskedCoupling's residual in its output stream
    (usf_conv2d_same_res_f32; transforms.py:277-306) -- one launch for what ConvSame + MaskedResidual do in two; the backward is
    theirs (d conv = sign * om * dy, d rx = dy).  ``ConvSameRes.served`` tells whether the kernel takes the shape."""

    @staticmethod
    def served(conv, x) -> bool:
        """the shapes of the fused form (include/usflows_hip.h); a launch that still declines falls back to the two passes"""
        return (config.conv_res and conv.kernel_size == (3, 3) and conv.in_channels in (16, 32) and conv.out_channels in (16, 32, 64)
                and x.shape[2] * x.shape[3] <= 64 and conv_shape_ok(conv, x.shape[0], x.shape[2], x.shape[3]))

    @staticmethod
    def forward(ctx, x, weight, bias, rx, om, sign):
        x, rx = x.contiguous(), rx.contiguous()
        w = weight.detach()
        ks = w.shape[2]
        planes, ctx.planes_t = _weight_planes(w, ctx.needs_input_grad[0], weight)
        b = None if bias is None else bias.detach().contiguous()
        y = _ext.conv2d_same_res(x, planes, w.shape[0], ks, rx, om, sign, bias=b)
        if y is None:
            y = _ext.masked_residual(rx, _ext.conv2d_same(x, planes, w.shape[0], ks, bias=b), om, sign)
        ctx.save_for_backward(x, w, om)
        ctx.cfg = (ks, bias is not None, sign)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, om = ctx.saved_tensors
        ks, has_bias, sign = ctx.cfg
        dy = dy.contiguous()
        dt = _ext.masked_residual(None, dy, om, sign)
        dW = db = dx = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            r = _ext.conv_wgrad(x, dt, ks, want_bias=has_bias, defer=_takeable(ctx.params),
                                owners=tuple(id(q) for q in ctx.params if q is not None))
            if r is None:
                raise RuntimeError("usflows_amd: usf_conv_wgrad_f32 does not serve this shape")
            dW, db = r
        if ctx.needs_input_grad[0]:
            planes_t = ctx.planes_t if ctx.planes_t is not None else _ext.conv2d_weight_planes(w, transposed=True)
            dx = _ext.conv2d_same(dt, planes_t, w.shape[1], ks)
        return dx, dW, db, dy, None, None


class ConvSameFork(torch.autograd.Function):
    """(conv(in_act(x) * in_mul) + bias, x): the convolution of a layer whose INPUT forks -- GatedConv (the skip connection,
    networks.py:108-122) and MaskedCoupling (the residual, transforms.py:277-306) use x a second time.  Returning x through the
    same node lets the backward see both gradients and write their sum from ONE pass: the data-gradient convolution adds the
    other branch's gradient in its output stream (usf_conv2d_same_gate_f32 with gate_add / usf_conv2d_same_res_f32) instead of
    autograd's separate accumulation pass over the batch"""

    @staticmethod
    def forward(ctx, x, weight, bias, in_mul, in_act):
        x = x.contiguous()
        w = weight.detach()
        ks = w.shape[2]
        ia = _act(in_act)
        planes, ctx.planes_t = _weight_planes(w, ctx.needs_input_grad[0], weight)
        y = _ext.conv2d_same(x, planes, w.shape[0], ks, bias=None if bias is None else bias.detach().contiguous(),
                             in_mul=in_mul, in_act=ia[0], in_slope=ia[1])
        ctx.save_for_backward(x, w, in_mul)
        ctx.cfg = (ks, in_act, bias is not None)
        ctx.params = (weight, bias)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dxs):
        x, w, in_mul = ctx.saved_tensors
        ks, in_act, has_bias = ctx.cfg
        dy = dy.contiguous()
        ia = _act(in_act)
        dW = db = dx = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            r = _ext.conv_wgrad(x, dy, ks, in_mul=in_mul, in_act=ia[0], in_slope=ia[1], want_bias=has_bias,
                                defer=_takeable(ctx.params), owners=tuple(id(q) for q in ctx.params if q is not None))
            if r is None:
                raise RuntimeError("usflows_amd: usf_conv_wgrad_f32 does not serve this shape")
            dW, db = r
        if ctx.needs_input_grad[0]:
            planes_t = ctx.planes_t if ctx.planes_t is not None else _ext.conv2d_weight_planes(w, transposed=True)
            dxs = None if dxs is None else dxs.contiguous()
            if dxs is not None and in_mul is not None and in_act is None:
                # dx = dxs + mask * dgrad: the residual form of the convolution
                dx = _ext.conv2d_same_res(dy, planes_t, w.shape[1], ks, dxs, in_mul, 1.0)
            elif dxs is not None and in_mul is None:
                dx = _ext.conv2d_same_gate(dy, planes_t, w.shape[1], ks, x, ia[1] if in_act is not None else 1.0, gate_add=dxs)
            if dx is None:
                if in_act is not None or in_mul is not None:
                    dx = _ext.conv2d_same_gate(dy, planes_t, w.shape[1], ks, x, ia[1] if in_act is not None else 1.0, in_mul)
                if dx is None:
                    dx = _ext.conv2d_same(dy, planes_t, w.shape[1], ks)
                    _gate_inplace(dx, x, in_act)
                    if in_mul is not None:
                        dx = _ext.masked_residual(None, dx, in_mul, 1.0)
                if dxs is not None:
                    dx = dx + dxs
        return dx, dW, db, None, None


class Pointwise(torch.autograd.Function):
    """bias + W in_act(x) per pixel on usf_pointwise_conv_f32 (W [cout, cin]: an nn.Conv2d weight with kernel 1)"""

    @staticmethod
    def forward(ctx, x, weight, bias, in_act):
        x = x.contiguous()
        w2 = weight.detach().reshape(weight.shape[0], weight.shape[1]).contiguous()
        ia = _act(in_act)
        y = _ext.pointwise_conv(x, w2, None if bias is None else bias.detach().contiguous(), in_act=ia[0], in_slope=ia[1])
        ctx.save_for_backward(x, w2)
        ctx.cfg = (in_act, bias is not None, tuple(weight.shape))
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        in_act, has_bias, wshape = ctx.cfg
        dy = dy.contiguous()
        ia = _act(in_act)
        dW = db = dx = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            r = _ext.conv_wgrad(x, dy, 1, in_act=ia[0], in_slope=ia[1], want_bias=has_bias, defer=_takeable(ctx.params), owners=tuple(id(q) for q in ctx.params if q is not None))
            if r is None:
                raise RuntimeError("usflows_amd: usf_conv_wgrad_f32 does not serve this shape")
            dW, db = r[0].reshape(wshape), r[1]
        if ctx.needs_input_grad[0]:
            if in_act is not None and ia[0] == _ext.ACT_LEAKY_RELU:
                # the derivative of the input's (Leaky)ReLU rides in the data-gradient pass
                dx = _ext.pointwise_conv(dy, w2.t().contiguous(), out_act=_ext.ACT_GATE, out_slope=ia[1], gate_x=x)
            else:
                dx = _ext.pointwise_conv(dy, w2.t().contiguous())
        return dx, dW, db, None


def pointwise_shape_ok(conv, B: int, H: int, W: int) -> bool:
    """a kernel-1 nn.Conv2d whose forward and data gradient run on usf_pointwise_conv_f32"""
    if not (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1):
        return False
    pad = conv.padding
    if not (pad == "same" or (not isinstance(pad, str) and tuple(pad) == (0, 0))):
        return False
    cin, cout = conv.in_channels, conv.out_channels
    return (_ext.pointwise_conv_supported(cin, cout, False) and _ext.pointwise_conv_supported(cout, cin, False)
            and _ext.load().usf_conv_wgrad_workspace(max(B, 1), cin, cout, H, W, 1) > 0)


class GatedResidual(torch.autograd.Function):
    """x + vg[:, :C] * sigmoid(vg[:, C:])  (GatedConv.forward's last line, networks.py:108-122)"""

    @staticmethod
    def forward(ctx, x, vg):
        x, vg = x.contiguous(), vg.contiguous()
        ctx.save_for_backward(vg)
        return _ext.gated_residual(x, vg)

    @staticmethod
    def backward(ctx, dy):
        (vg,) = ctx.saved_tensors
        dy = dy.contiguous()
        return dy, _ext.gated_residual_bwd(dy, vg)




def gated_tail_ok(conv2, x, ln=None) -> bool:
    """GatedConv's 1 x 1 convolution `conv2` (C -> 2 C) with the gate, the skip connection [and the (Leaky)ReLU + LayerNormChannels
    that follow] as ONE differentiable launch each way (usf_gated_tail_f32 / usf_gated_tail_bwd_f32) at this input"""
    if not (config.gated_tail and isinstance(conv2, torch.nn.Conv2d) and conv2.kernel_size == (1, 1) and conv2.stride == (1, 1)
            and conv2.groups == 1 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    pad = conv2.padding
    if not (pad == "same" or (not isinstance(pad, str) and tuple(pad) == (0, 0))):
        return False
    B, Cc, H, W = x.shape
    if not (conv2.in_channels == Cc and conv2.out_channels == 2 * Cc and 0 < B * H * W <= config.gated_tail_max_pixels
            and _ext.gated_tail_supported(Cc)):
        return False
    if ln is not None and not (ln.gamma.numel() == Cc and ln.gamma.dtype == torch.float32):
        return False
    return _ext.load().usf_conv_wgrad_workspace(max(B, 1), Cc, 2 * Cc, H, W, 1) > 0


class GatedTail(torch.autograd.Function):
    """LayerNormChannels(post_act(x + val * sigmoid(gate))), [val, gate] = W in_act(h) + bias -- the tail of a GatedConv layer
    of ConvNet2D (networks.py:108-122, 40-58, 480-493) on usf_gated_tail_f32; gamma None: no layer norm, y = x + val * sigmoid(gate).
    The backward recomputes val / gate from (h, x): ONE launch for dx, dh and the partial sums of dW, dbias, dgamma, dbeta."""

    @staticmethod
    def forward(ctx, h, x, weight, bias, gamma, beta, in_act, post_act, eps):
        h, x = h.contiguous(), x.contiguous()
        w2 = weight.detach().reshape(weight.shape[0], weight.shape[1]).contiguous()
        b2 = None if bias is None else bias.detach().contiguous()
        ia, pa = _act(in_act), _act(post_act)
        ln = None
        if gamma is not None:
            ln = (gamma.detach().reshape(-1).contiguous(), beta.detach().reshape(-1).contiguous(), float(eps))
        y = _ext.gated_tail(h, x, w2, b2, ia[0], ia[1], pa[0], pa[1], ln)
        ctx.save_for_backward(h, x, w2, b2, *(ln[:2] if ln is not None else ()))
        ctx.cfg = (in_act, post_act, float(eps), tuple(weight.shape), bias is not None, None if gamma is None else tuple(gamma.shape))
        ctx.params = (weight, bias, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        in_act, post_act, eps, wshape, has_bias, pshape = ctx.cfg
        saved = ctx.saved_tensors
        h, x, w2, b2 = saved[:4]
        ln = (saved[4], saved[5], eps) if pshape is not None else None
        ia, pa = _act(in_act), _act(post_act)
        weight, bias, gamma, beta = ctx.params
        params = (weight, bias, gamma, beta)
        dx, dh, dW, db, dg, dbt, _ = _ext.gated_tail_bwd(h, x, dy.contiguous(), w2, b2, ia[0], ia[1], pa[0], pa[1], ln,
                                                         defer=_takeable(params), owners=tuple(id(q) for q in params if q is not None))
        dW = dW.view(wshape)
        if not has_bias:
            db = None
        if pshape is not None:
            dg, dbt = dg.view(pshape), dbt.view(pshape)
        return dh, dx, dW, db, dg, dbt, None, None, None


class LayerNormCh(torch.autograd.Function):
    """LayerNormChannels (networks.py:40-58) with the (Leaky)ReLU ConvNet2D puts in front of it"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, pre_act):
        x = x.contiguous()
        g = gamma.detach().reshape(-1).contiguous()
        a = _act(pre_act)
        y = _ext.layernorm_channels(x, g, beta.detach().reshape(-1).contiguous(), eps, a[0], a[1])
        ctx.save_for_backward(x, g)
        ctx.cfg = (eps, pre_act, tuple(gamma.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        eps, pre_act, pshape = ctx.cfg
        a = _act(pre_act)
        dx, dg, dbt = _ext.layernorm_channels_bwd(x, dy.contiguous(), g, eps, a[0], a[1])
        return dx, dg.reshape(pshape), dbt.reshape(pshape), None, None


class MaskedResidual(torch.autograd.Function):
    """x + sign * (1 - mask) * t  (MaskedCoupling on image-shaped inputs, transforms.py:277-306)"""

    @staticmethod
    def forward(ctx, x, t, one_minus_mask, sign):
        ctx.save_for_backward(one_minus_mask)
        ctx.sign = sign
        return _ext.masked_residual(x.contiguous(), t.contiguous(), one_minus_mask, sign)

    @staticmethod
    def backward(ctx, dy):
        (om,) = ctx.saved_tensors
        dy = dy.contiguous()
        dt = _ext.masked_residual(None, dy, om, ctx.sign) if ctx.needs_input_grad[1] else None
        return dy, dt, None, None


class ChannelAffine(torch.autograd.Function):
    """the 1 x 1 convolution of BlockAffineTransform on NCHW data: pre_sub False: y = W x + b (forward direction);
    pre_sub True: y = W (x - b) (backward direction, W = M^-1); W [C, C], b [C] are tiny differentiable torch tensors"""

    @staticmethod
    def forward(ctx, x, W, b, pre_sub, Wt=None, defer=False):
        """Wt: W^T, contiguous (the data gradient's map; made here when not given).  defer: the consumer of dW / db is a node
        that issues queued sums before it reads them (RunsOut / AffinePrep) -- the weight gradient's last sum may be queued"""
        x = x.contiguous()
        Wd, bd = W.detach().contiguous(), b.detach().contiguous()
        y = torch.empty_like(x)
        if pre_sub:
            _ext.channel_affine(x, y, Wd, pre_sub=bd)
        else:
            _ext.channel_affine(x, y, Wd, bias=bd)
        ctx.save_for_backward(x, Wd, bd, Wt)
        ctx.pre_sub = pre_sub
        ctx.defer = bool(defer) and not pre_sub
        return y

    @staticmethod
    def backward(ctx, dy):
        x, Wd, bd, Wt = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dW = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            r = _ext.conv_wgrad(x, dy, 1, pre_sub=bd if ctx.pre_sub else None, want_bias=True, defer=ctx.defer, owners=(id(ctx),))
            if r is None:
                raise RuntimeError("usflows_amd: usf_conv_wgrad_f32 does not serve this channel count")
            dW = r[0].reshape(Wd.shape)
            db = -(Wd.t() @ r[1]) if ctx.pre_sub else r[1]
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(dy)
            _ext.channel_affine(dy, dx, Wt if Wt is not None else Wd.t().contiguous())
        return dx, dW, db, None, None, None


def channel_affine_train_ok(x, C) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == C and C in (16, 32, 48, 64)
            and _ext.load().usf_conv_wgrad_workspace(max(x.shape[0], 1), C, C, x.shape[2], x.shape[3], 1) > 0)


class BaseLogProb(torch.autograd.Function):
    """sum_d log p(z_d) of a Laplace / Normal base with fixed parameters (usf_base_logprob_f32 and its gradient kernel)"""

    @staticmethod
    def forward(ctx, z, loc, scale, base_id):
        B = z.shape[0]
        D = loc.numel()
        zf = z.reshape(B, D).contiguous()
        out = torch.empty(B, dtype=torch.float32, device=z.device)
        _ext.base_logprob(zf, D, B, D, base_id, loc, scale, 0.0, out)
        ctx.save_for_backward(zf, loc, scale)
        ctx.cfg = (base_id, tuple(z.shape))
        return out

    @staticmethod
    def backward(ctx, g_lp):
        zf, loc, scale = ctx.saved_tensors
        base_id, shape = ctx.cfg
        B, D = zf.shape
        g = torch.empty_like(zf)
        _ext.base_logprob_grad(zf, D, g_lp.contiguous(), B, D, base_id, loc, scale, g, D)
        return g.reshape(shape), None, None, None


def needs_grad(module, *tensors) -> bool:
    if not torch.is_grad_enabled():
        return False
    if any(torch.is_tensor(t) and t.requires_grad for t in tensors):
        return True
    return any(p.requires_grad for p in module.parameters())


# ---- the affine blocks' parameter maps, batched over the blocks of a flow ---------------------------------------------------
# The reference evaluates matrix() / inverse_matrix() / bias() / log_abs_det_jacobian() of every BlockAffineTransform with a
# few dozen torch ops on C x C tensors each (transforms.py:1283-1320, 1457-1476), once per use: with affine conjugation a
# training step of the 2-block MNIST model is ~800 kernel launches of 3-4 us of which ~40 touch the batch.  Under
# ``batched_affine_prep`` the same maps are evaluated ONCE per log_prob call for all blocks of equal structure as [n, C, C]
# stacks (same formulas, torch autograd differentiates them); the layers pick their (M, M^-1, b, log|det|) from the stack.
class _PrepState(threading.local):
    """the current pass's prep, per thread (two threads training two flows must not see each other's maps)"""

    def __init__(self):
        self.prep = None       # id(block transform) -> (M, Minv, b, ladj[, c, (group, row)])
        self.stacks = {}       # group -> stacked log|det| [n] of the current pass (device prep only)
        self.maps = {}         # group -> ([M | M^-1] [2n, C, C], [b | c] [2n, C], n) of the current pass (device prep only)
        self.runs = None       # flows.Flow._train_affine_run: the composed runs of the current pass


_STATE = _PrepState()


def current_prep(block_transform):
    """(M, Minv, b, ladj[, c = -Minv b, (group, row)]) of this block transform inside ``batched_affine_prep``; None outside
    or when not covered"""
    return None if _STATE.prep is None else _STATE.prep.get(id(block_transform))


_EYES = {}


def _eye(C: int, device) -> torch.Tensor:
    """the C x C identity on this device, made once (a constant: every call of the batched prep would otherwise spend two
    launches on it)"""
    key = (C, str(device))
    e = _EYES.get(key)
    if e is None:
        if torch.cuda.is_available() and torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
            return torch.eye(C, dtype=torch.float32, device=device)      # (never fill a cache inside a stream capture)
        e = _EYES[key] = torch.eye(C, dtype=torch.float32, device=device)
    return e


def _parts(bt):
    from . import transforms as T
    parts = list(bt.transforms) if isinstance(bt, T.SequentialAffineTransform) else [bt]
    sig = []
    for t in parts:
        if isinstance(t, T.LUTransform):
            sig.append(("lu", t.dim))
        elif isinstance(t, T.HouseholderTransform):
            sig.append(("hh", t.dim, t.nvs))
        else:
            return None, None
    return parts, tuple(sig)


class AffinePrep(torch.autograd.Function):
    """(M, M^-1, b, log|det|) of n affine block transforms of equal structure -- LUTransform [+ HouseholderTransform] --
    as ONE launch (usf_affine_prep_f32), and the gradients of their parameters as one more (usf_affine_prep_bwd_f32):
    the formulas of ``_prep_group`` below and what autograd derives from them.  Inputs are the stacked parameters; the 4 n
    outputs are handed out per block (views of the stacked results), so that the layers' gradients arrive per block and
    are gathered by one ``stack`` per kind instead of a select-backward + add chain per use."""

    @staticmethod
    def forward(ctx, Lr, Ur, bias, vk, w0):
        M, Minv, b, c, ladj, save = _ext.affine_prep(Lr, Ur, bias, vk, w0)
        ctx.save_for_backward(save, bias, vk, w0, Minv, b)
        # (5 n + 3 outputs, most of them unused in a given pass: no zero tensors -- a fill launch each -- for those)
        ctx.set_materialize_grads(False)
        n = Lr.shape[0]
        ctx.n, ctx.C = n, Lr.shape[1]
        # per block: M, M^-1, b, c = -M^-1 b, log|det|; and once more the stacked log|det| (the flow's total log-det is one
        # weighted sum over it: Flow._layer_loop_log_prob)
        # ... and the stacked [M | M^-1] and [b | c] (compose_runs gathers the maps of a run's layers from them)
        return tuple(M.unbind(0)) + tuple(Minv.unbind(0)) + tuple(b.unbind(0)) + tuple(c.unbind(0)) + tuple(ladj.unbind(0)) \
            + (ladj.view(n), M._base.view(2 * n, ctx.C, ctx.C), b._base.view(2 * n, ctx.C))

    @staticmethod
    def backward(ctx, *grads):
        save, bias, vk, w0, Minv, b = ctx.saved_tensors
        n, C = ctx.n, ctx.C
        dev = save.device
        # (gradients of single affine layers may still be waiting for their last sum: image_training.ChannelAffine)
        _ext.flush_partial_sums(torch._C._current_graph_task_id(), final=False)

        def gather(gs, shape):
            if all(g is None for g in gs):
                return _zeros((n,) + shape, dev)
            z = _zeros(shape, dev)
            return torch.stack([z if g is None else g for g in gs]).contiguous()

        dM = gather(grads[0:n], (C, C))
        dMinv = gather(grads[n:2 * n], (C, C))
        db = gather(grads[2 * n:3 * n], (C,))
        dc = gather(grads[3 * n:4 * n], (C,))
        dl_each, dl_all = grads[4 * n:5 * n], grads[5 * n]
        if dl_all is None and all(g is None for g in dl_each):
            dl_all = _zeros((n,), dev)
        gMM, gbc = grads[5 * n + 1], grads[5 * n + 2]
        if gMM is not None:
            dM, dMinv = dM + gMM[:n], dMinv + gMM[n:]
        if gbc is not None:
            db, dc = db + gbc[:n], dc + gbc[n:]
        if dl_all is not None and all(g is None for g in dl_each):
            dl = dl_all.contiguous()
        else:
            dl = gather(dl_each, ())
            if dl_all is not None:
                dl = dl + dl_all
        dLr, dUr, dbias, dvk = _ext.affine_prep_bwd(save, bias, vk, w0, Minv, b, dM, dMinv, db, dc, dl)
        # (the kernel writes zeros outside L's / U's triangle: LUTransform's gradient hooks need not multiply by their masks)
        mark_masked(dLr)
        mark_masked(dUr)
        return dLr, dUr, dbias, dvk, None


# ---- gradients that already carry LUTransform's structural zeros ------------------------------------------------------------------
# LUTransform masks the gradients of L_raw / U_raw with a tensor hook each (transforms.py:563-564 here, the reference's
# LUTransform likewise): 2 launches per block and pass.  usf_affine_prep_bwd_f32 writes exact zeros outside the triangles, so
# for ITS rows the product changes nothing; AffinePrep.backward registers the rows it is about to hand to autograd (address +
# shape, per backward pass), the hook asks ``take_masked(grad)`` and skips the product for exactly those tensors, once each.
_masked_lock = threading.Lock()
_masked = {}          # autograd graph task id -> {(data_ptr, shape)}


def mark_masked(stack) -> None:
    """stack: a [n, D, D] tensor (its rows are registered) or a list of tensors"""
    task = torch._C._current_graph_task_id()
    if task < 0:
        return
    with _masked_lock:
        for old in [t for t in _masked if t < task - 256]:
            del _masked[old]
        s = _masked.setdefault(task, set())
        for row in (stack.unbind(0) if torch.is_tensor(stack) else stack):
            s.add((row.data_ptr(), tuple(row.shape)))


def take_masked(grad: torch.Tensor) -> bool:
    task = torch._C._current_graph_task_id()
    if task < 0 or not _masked:
        return False
    key = (grad.data_ptr(), tuple(grad.shape))
    with _masked_lock:
        s = _masked.get(task)
        if s is not None and key in s and grad.is_contiguous():
            s.discard(key)
            return True
    return False


_ZEROS = {}


def _zeros(shape, device) -> torch.Tensor:
    """a constant zero tensor of this shape (read-only: stands in for gradients that did not arrive)"""
    key = (tuple(shape), str(device))
    z = _ZEROS.get(key)
    if z is None:
        if torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing():
            return torch.zeros(shape, dtype=torch.float32, device=device)      # (never fill a cache inside a stream capture)
        z = _ZEROS[key] = torch.zeros(shape, dtype=torch.float32, device=device)
    return z


_COEFS = {}


def coef_tensor(values, device) -> torch.Tensor:
    """a constant fp32 vector with these values on the device, uploaded once (the log-det weights of a flow's layers)"""
    key = (tuple(values), str(device))
    t = _COEFS.get(key)
    if t is None:
        t = torch.tensor(list(values), dtype=torch.float32, device=device)
        if not (torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing()):
            _COEFS[key] = t
    return t


def prep_stack(group):
    return _STATE.stacks.get(group)


affine_prep_kernels = True        # USFLOWS_AMD_AFFINE_PREP=0 / False: the batched torch formulation below


def _prep_group_device(bts, parts_list, sig, device):
    """the group's maps through usf_affine_prep_f32, or None when the structure is not the kernel's (one LUTransform,
    optionally followed by one HouseholderTransform; C <= 64; fp32 parameters on a GPU)"""
    import os
    if not (affine_prep_kernels and config.affine_prep):
        return None
    if torch.device(device).type != "cuda" or tuple(k[0] for k in sig) not in (("lu",), ("lu", "hh")):
        return None
    C = sig[0][1]
    if C > _ext.AFFINE_PREP_MAX_C or (len(sig) == 2 and (sig[1][1] != C or not 1 <= sig[1][2] <= 8)):
        return None
    lus = [p[0] for p in parts_list]
    if any(t.dtype != torch.float32 for m in lus for t in (m.L_raw, m.U_raw, m.bias_vector)):
        return None
    _ext.load()
    n = len(bts)
    Lr = torch.stack([m.L_raw for m in lus])
    Ur = torch.stack([m.U_raw for m in lus])
    bias = torch.stack([m.bias_vector for m in lus])
    vk = w0 = None
    if len(sig) == 2:
        hhs = [p[1] for p in parts_list]
        if any(h.vk_householder.dtype != torch.float32 for h in hhs):
            return None
        vk = torch.stack([h.vk_householder for h in hhs])
        w0 = torch.stack([h.w_0.detach() for h in hhs])
    out = AffinePrep.apply(Lr, Ur, bias, vk, w0)
    key = (sig, tuple(id(bt) for bt in bts))
    _STATE.stacks[key] = out[5 * n]
    _STATE.maps[key] = (out[5 * n + 1], out[5 * n + 2], n)
    # (M, M^-1, b, log|det|, c = -M^-1 b, (group, row) of the block in the stacked log|det|)
    return {id(bt): (out[i], out[n + i], out[2 * n + i], out[4 * n + i], out[3 * n + i], (key, i)) for i, bt in enumerate(bts)}


def _prep_group(bts, parts_list, sig, device):
    dev_out = _prep_group_device(bts, parts_list, sig, device)
    if dev_out is not None:
        return dev_out
    n, C = len(bts), sig[0][1]
    eye = _eye(C, device)
    M = Minv = b = None
    ladj = torch.zeros(n, dtype=torch.float32, device=device)
    for pos, kind in enumerate(sig):
        mods = [p[pos] for p in parts_list]
        if kind[0] == "lu":
            L = torch.stack([m.L_raw for m in mods]).tril(-1) + eye
            U = torch.stack([m.U_raw for m in mods]).triu()
            Mj = torch.matmul(L, U)
            eye_n = eye.expand(n, C, C)
            Mij = torch.matmul(torch.linalg.solve_triangular(U, eye_n, upper=True), torch.linalg.solve_triangular(L, eye_n, upper=False))
            bj = torch.stack([m.bias_vector for m in mods])
            ladj = ladj + U.diagonal(dim1=-2, dim2=-1).abs().log().sum(-1)
        else:
            v = torch.stack([m.vk_householder for m in mods])                # [n, nvs, C]
            Mj = torch.stack([m.w_0 for m in mods])
            for k in range(kind[2]):
                vk = v[:, k]
                Mj = torch.matmul(Mj, eye - 2 * vk.unsqueeze(2) * vk.unsqueeze(1) / (vk * vk).sum(-1).view(n, 1, 1))
            Mij = Mj.transpose(1, 2)
            bj = None
        # SequentialAffineTransform (transforms.py:1457-1476): matrix = M_1 M_2 ..., inverse = M_k^-1 ... M_1^-1, b <- b M_j + b_j
        if M is None:
            M, Minv, b = Mj, Mij, (bj if bj is not None else torch.zeros(n, C, dtype=torch.float32, device=device))
        else:
            M = torch.matmul(M, Mj)
            Minv = torch.matmul(Mij, Minv)
            b = torch.matmul(b.unsqueeze(1), Mj).squeeze(1)
            if bj is not None:
                b = b + bj
    return {id(bt): (M[i], Minv[i], b[i], ladj[i]) for i, bt in enumerate(bts)}


class batched_affine_prep:
    """context manager around one log_prob / forward pass of an image-shaped flow in training (see above)"""

    def __init__(self, layers, device):
        self.layers, self.device, self.prev = layers, device, None

    def __enter__(self):
        from . import transforms as T
        self.prev = _STATE.prep
        groups = {}
        for l in self.layers:
            blk = l.transform if isinstance(l, T.InverseTransform) else l
            if not (isinstance(blk, T.BlockAffineTransform) and blk.input_rank >= 1):
                continue
            bt = blk.block_transform
            parts, sig = _parts(bt)
            if parts is None or any(p.device != self.device for p in bt.parameters()):
                continue
            g = groups.setdefault(sig, ({}, []))
            if id(bt) not in g[0]:
                g[0][id(bt)] = bt
                g[1].append(parts)
        prep = {}
        self.prev_stacks = (_STATE.stacks, _STATE.maps, _STATE.runs)
        _STATE.stacks, _STATE.maps, _STATE.runs = {}, {}, None
        for sig, (bts, parts_list) in groups.items():
            prep.update(_prep_group(list(bts.values()), parts_list, sig, self.device))
        _STATE.prep = prep
        return self

    def __exit__(self, *exc):
        _STATE.prep = self.prev
        _STATE.stacks, _STATE.maps, _STATE.runs = self.prev_stacks
        return False


# ---- runs of consecutive affine layers, composed for ALL runs of a pass at once -------------------------------------------------
# With affine_conjugation a coupling is followed by block_i^-1 and block_(i+1) (flows.py:452-470): y = A2 (A1 x + c1) + c2 is one
# channel-affine pass on A = A2 A1, c = A2 c1 + c2.  Composing run by run is three tiny torch launches per run and six in the
# backward pass (14 runs in the live MNIST configuration: ~125 of the ~1 000 launches of its training step at batch 32); here
# the runs of equal length and direction pattern are composed together: one index_select per position from the prep kernel's
# stacked maps and one batched product per position -- a dozen launches per pass, whatever the number of runs.
_IDX = {}


def _onehot(rows, width, device) -> torch.Tensor:
    """the constant [len(rows), width] fp32 matrix with a one at (i, rows[i]), uploaded once"""
    key = (tuple(rows), width, str(device))
    t = _IDX.get(key)
    if t is None:
        host = torch.zeros(len(rows), width, dtype=torch.float32)
        host[torch.arange(len(rows)), torch.tensor(list(rows))] = 1.0
        t = host.to(device)
        if not (torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing()):
            _IDX[key] = t
    return t


class RunsOut(torch.autograd.Function):
    """the composed maps handed out per run; the backward first issues the sums the runs' weight gradients queued
    (ChannelAffine with defer=True), then gathers the gradients with one ``stack`` per kind"""

    @staticmethod
    def forward(ctx, A, cv):
        ctx.R, ctx.C = A.shape[0], A.shape[1]
        return tuple(A.unbind(0)) + tuple(cv.unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        R, C = ctx.R, ctx.C
        _ext.flush_partial_sums(torch._C._current_graph_task_id(), final=False)
        dev = next(g.device for g in grads if g is not None)

        def gather(gs, shape):
            if all(g is None for g in gs):
                return None
            z = _zeros(shape, dev)
            return torch.stack([z if g is None else g for g in gs])

        return gather(grads[:R], (C, C)), gather(grads[R:], (C,))


def compose_runs(specs):
    """specs: [(group, [(row, inv), ...])] -- runs of >= 2 affine layers whose maps are rows of the group's prep stacks (inv: the
    layer is an InverseTransform, i.e. its backward is the block's forward map M, b; else M^-1, c = -M^-1 b).  Returns
    [(A, c, A^T)] per run: y = A x + c for the run applied in order, A^T a contiguous copy for the data gradient."""
    out = [None] * len(specs)
    classes = {}
    for r, (group, rows) in enumerate(specs):
        classes.setdefault((group, len(rows), tuple(inv for _, inv in rows)), []).append(r)
    for (group, L, pattern), members in classes.items():
        MM, bc, n = _STATE.maps[group]
        dev = MM.device
        A = cv = None
        C = MM.shape[1]
        for j in range(L):
            # (rows picked by a constant one-hot matrix: exact -- 1.0 * x plus zeros -- and its backward is one small product
            # instead of index_select's zero-fill + index_add, 40 us per stack at these sizes)
            S = _onehot([specs[r][1][j][0] + (0 if pattern[j] else n) for r in members], 2 * n, dev)
            Aj, cj = torch.mm(S, MM.view(2 * n, C * C)).view(len(members), C, C), torch.mm(S, bc)
            if A is None:
                A, cv = Aj, cj
            else:
                cv = torch.baddbmm(cj.unsqueeze(2), Aj, cv.unsqueeze(2)).squeeze(2)
                A = torch.bmm(Aj, A)
        At = A.detach().transpose(1, 2).contiguous()
        outs = RunsOut.apply(A, cv)
        R = len(members)
        for i, r in enumerate(members):
            out[r] = (outs[i], outs[R + i], At[i])
    return out
