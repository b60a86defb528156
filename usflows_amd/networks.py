"""Conditioner networks of the hot path (flat inputs): dense MLPs.

``ConditionalDenseNN`` mirrors the reference's in-repo class (networks.py:681-751: same ctor
arguments, ``layers`` ModuleList ordering [input, context, hidden..., output] and therefore the
same state-dict keys).  ``DenseNN`` stands in for ``pyro.nn.DenseNN`` (pyro-ppl 1.8.6), which the
live vector configs name by dotted path (experiments/synthetic/gaussian_mixture.yaml:66-71);
pyro is not a dependency of this package.  ``ConvNet`` mirrors the *vector* path of the reference's
generic ``ConvNet`` (networks.py:246-308, forward 379-389; ``in_dims = [D]``): with ``gating=False,
normalize_layers=False`` it is a piece-wise linear MLP and runs on the fused device path; the gated /
layer-normalised defaults are mirrored for the torch path only (they are not piece-wise linear).
The spatial (CNN) path is out of scope.

The modules themselves are plain torch (used under autograd / on CPU); on the device fast path
``usflows_amd.engine`` reads their parameters and runs the whole MLP inside the fused coupling
kernel -- ``forward`` below is not called there.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch

from .config import config  # noqa: E402
from torch import nn


class ConditionalDenseNN(nn.Module):
    def __init__(self, input_dim, context_dim, hidden_dims, out_dim, nonlinearity=nn.ReLU()):
        super().__init__()
        self.input_dim = input_dim
        self.context_dim = context_dim
        self.hidden_dims = hidden_dims
        self.out_dim = out_dim
        layers = [nn.Linear(input_dim, hidden_dims[0]), nn.Linear(context_dim, hidden_dims[0])]
        for i in range(1, len(hidden_dims)):
            layers.append(nn.Linear(hidden_dims[i - 1], hidden_dims[i]))
        layers.append(nn.Linear(hidden_dims[-1], out_dim))
        self.layers = nn.ModuleList(layers)
        self.f = nonlinearity

    def forward(self, x, context=None):
        h = self.layers[0](x)
        if context is not None:
            h = h + self.layers[1](context)
        h = self.f(h)
        for layer in self.layers[2:-1]:
            h = self.f(layer(h))
        return self.layers[-1](h)


class DenseNN(nn.Module):
    """Plain MLP with pyro's ``DenseNN`` constructor: Linear->f->...->Linear(sum(param_dims)),
    no activation on the output; one tensor when ``len(param_dims) == 1``, else a tuple of slices."""

    def __init__(self, input_dim: int, hidden_dims: Sequence[int], param_dims: Sequence[int] = (1, 1),
                 nonlinearity: nn.Module = nn.ReLU()):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dims = list(hidden_dims)
        self.param_dims = list(param_dims)
        self.count_params = len(self.param_dims)
        self.output_multiplier = sum(self.param_dims)
        ends = torch.cumsum(torch.tensor(self.param_dims), dim=0)
        starts = torch.cat((torch.zeros(1).type_as(ends), ends[:-1]))
        self.param_slices = [slice(int(s), int(e)) for s, e in zip(starts, ends)]
        layers = [nn.Linear(input_dim, self.hidden_dims[0])]
        for i in range(1, len(self.hidden_dims)):
            layers.append(nn.Linear(self.hidden_dims[i - 1], self.hidden_dims[i]))
        layers.append(nn.Linear(self.hidden_dims[-1], self.output_multiplier))
        self.layers = nn.ModuleList(layers)
        self.f = nonlinearity

    def forward(self, x):
        h = x
        for layer in self.layers[:-1]:
            h = self.f(layer(h))
        h = self.layers[-1](h)
        if self.output_multiplier == 1:
            return h
        h = h.reshape(list(x.size()[:-1]) + [self.output_multiplier])
        if self.count_params == 1:
            return h
        return tuple(h[..., s] for s in self.param_slices)


class LayerNormVector(nn.Module):
    """LayerNorm over the feature axis of (batch, features) inputs (networks.py:206-219)."""

    def __init__(self, features: int, eps: float = 1e-5):
        super().__init__()
        self.layernorm = nn.LayerNorm(features, eps=eps)

    def forward(self, x):
        return self.layernorm(_as_batch_features(x))


class GatedMLP(nn.Module):
    """Gated residual block of the vector ConvNet (networks.py:222-245):
    ``x' + val * sigmoid(gate)`` with ``[val, gate] = Linear(f(Linear(f(x))))`` and ``x'`` = x or a projection."""

    def __init__(self, in_features: int, out_features: int, nonlinearity=nn.ReLU()):
        super().__init__()
        self.net1 = nn.Sequential(nonlinearity, nn.Linear(in_features, out_features), nonlinearity,
                                  nn.Linear(out_features, 2 * out_features))
        self.proj = nn.Linear(in_features, out_features) if in_features != out_features else None

    def forward(self, x):
        val, gate = self.net1(x).chunk(2, dim=1)
        skip = x if self.proj is None else self.proj(x)
        return skip + val * torch.sigmoid(gate)


def _as_batch_features(x):
    """(batch, features, 1) and (1, batch, features) are accepted as (batch, features) (networks.py:379-387)."""
    if x.dim() == 3 and x.shape[-1] == 1:
        x = x.view(x.shape[0], x.shape[1])
    if x.dim() == 3 and x.shape[0] == 1 and x.shape[2] != 1:
        x = x.permute(1, 2, 0).contiguous().view(x.shape[1], x.shape[2])
    return x


class _ConvStack(nn.Module):
    """a ``self.nn`` sequence of Conv2d / GatedConv / nonlinearity / LayerNormChannels modules on [B, C, H, W] inputs run as
    device passes (inference and training): shared by ``ConvNet2D`` and the 2-D spatial path of ``ConvNet``"""

    def forward(self, x, context=None, in_mul=None, residual=None):
        """in_mul (device path only, see ``first_conv_on_device``): a [C * H * W] mask the FIRST convolution multiplies
        into its input -- MaskedCoupling hands over the unmasked x and its mask instead of a masked copy.
        residual = (res_x, one_minus_mask, sign) (device path only): the caller is a MaskedCoupling that wants
        res_x + sign * one_minus_mask * net(x); the return value is then (tensor, done) -- done: the LAST convolution
        wrote the coupling's output itself (usf_conv2d_same_res_f32), otherwise tensor is net(x) as usual"""
        mods = list(self.nn)
        if not (x.is_cuda and x.dtype == torch.float32):
            assert in_mul is None and residual is None
            return self.nn(x)
        if residual is None and self.train_on_device(x):
            return self._forward_train_device(x, in_mul)
        done = False
        # the same module sequence on the device: convolutions on usf_conv2d_same_f32 (a nonlinearity behind a plain
        # convolution rides in its epilogue), a (Leaky)ReLU in front of a LayerNormChannels joins that layer's pass
        k = 0
        while k < len(mods):
            m = mods[k]
            nxt = mods[k + 1] if k + 1 < len(mods) else None
            act = _relu_kind(m)
            if isinstance(m, nn.Conv2d) and _conv_hip_ok(m, x):
                fold = _relu_kind(nxt) if nxt is not None else None
                after = mods[k + 2] if k + 2 < len(mods) else None
                if fold is not None and isinstance(after, LayerNormChannels):
                    fold = None                                   # that ReLU belongs to the layer norm's pass
                if residual is not None and k == len(mods) - 1 and fold is None:
                    x, done = _conv_hip(m, x, in_mul=in_mul if k == 0 else None, residual=residual)
                else:
                    x = _conv_hip(m, x, in_mul=in_mul if k == 0 else None, out_act=fold)
                k += 2 if fold is not None else 1
            elif isinstance(m, GatedConv) and _relu_kind(nxt) is not None and k + 2 < len(mods) \
                    and m.can_join_layernorm(x, mods[k + 2]):
                x = m(x, post=(_relu_kind(nxt), mods[k + 2]))     # GatedConv + nonlinearity + layer norm: its last pass does all
                k += 3
            elif act is not None and isinstance(nxt, LayerNormChannels) and nxt._hip_ok(x):
                x = nxt(x, pre_act=act)
                k += 2
            else:
                assert not (k == 0 and in_mul is not None)
                x = m(x)
                k += 1
        return (x, done) if residual is not None else x

    # ---- training on the device (rows N2 x N4): the same module sequence as differentiable device passes ----------------
    def train_on_device(self, x) -> bool:
        """True when this call needs gradients and every module of the sequence has a device backward at this shape
        (image_training.py); USFLOWS_AMD_IMAGE_TRAIN=0 keeps torch autograd"""
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] > 0
                and torch.is_grad_enabled() and config.image_train):
            return False
        if not (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False
        from . import image_training as it
        B, C, H, W = x.shape
        mods = list(self.nn)
        k = 0
        while k < len(mods):
            m = mods[k]
            nxt = mods[k + 1] if k + 1 < len(mods) else None
            if isinstance(m, nn.Conv2d):
                if m.in_channels != C or not it.conv_shape_ok(m, B, H, W):
                    return False
                C = m.out_channels
                after = mods[k + 2] if k + 2 < len(mods) else None
                k += 2 if (_relu_kind(nxt) is not None and not isinstance(after, LayerNormChannels)) else 1
            elif isinstance(m, GatedConv):
                if not m.train_shape_ok(B, C, H, W):
                    return False
                k += 1
            elif _relu_kind(m) is not None and isinstance(nxt, LayerNormChannels):
                if nxt.gamma.numel() != C or C > 64:
                    return False
                k += 2
            elif isinstance(m, LayerNormChannels):
                if m.gamma.numel() != C or C > 64:
                    return False
                k += 1
            elif _relu_kind(m) is not None:
                k += 1                                            # (a stand-alone nonlinearity: a torch elementwise op)
            else:
                return False
        return True

    def _forward_train_device(self, x, in_mul=None, fork: bool = False, residual=None):
        """fork (a MaskedCoupling caller that uses x again for its residual): returns (net(x), x') where x' is x passed through
        the first convolution's autograd node -- see image_training.ConvSameFork.  residual = (one_minus_mask, sign) with fork:
        the coupling's output x' + sign * (1 - mask) * net(x) may leave the LAST convolution's launch
        (image_training.ConvSameRes); returns (output, x', done)"""
        from . import image_training as it
        mods = list(self.nn)
        assert in_mul is None or isinstance(mods[0], nn.Conv2d)
        x_in = x
        x_fork = None
        k = 0
        while k < len(mods):
            m = mods[k]
            nxt = mods[k + 1] if k + 1 < len(mods) else None
            if isinstance(m, nn.Conv2d):
                fold = _relu_kind(nxt) if nxt is not None else None
                after = mods[k + 2] if k + 2 < len(mods) else None
                if fold is not None and isinstance(after, LayerNormChannels):
                    fold = None                                   # that ReLU belongs to the layer norm's pass
                if fork and k == 0 and fold is None:
                    x, x_fork = it.ConvSameFork.apply(x, m.weight, m.bias, in_mul, None)
                    k += 1
                    continue
                if residual is not None and k > 0 and k == len(mods) - 1 and m.out_channels == x_in.shape[1] \
                        and it.ConvSameRes.served(m, x):
                    y = it.ConvSameRes.apply(x, m.weight, m.bias, x_fork if x_fork is not None else x_in, residual[0], residual[1])
                    return y, x_fork, True
                x = it.ConvSame.apply(x, m.weight, m.bias, in_mul if k == 0 else None, None, fold)
                k += 2 if fold is not None else 1
            elif isinstance(m, GatedConv):
                # (few pixels: the nonlinearity and layer norm behind the block join its tail's launch, each way)
                after = mods[k + 2] if k + 2 < len(mods) else None
                if isinstance(nxt, LayerNormChannels) and it.gated_tail_ok(m.net[3], x, nxt):
                    x = m.forward_train_device(x, post=(None, nxt))
                    k += 2
                elif nxt is not None and _relu_kind(nxt) is not None and isinstance(after, LayerNormChannels) \
                        and it.gated_tail_ok(m.net[3], x, after):
                    x = m.forward_train_device(x, post=(_relu_kind(nxt), after))
                    k += 3
                else:
                    x = m.forward_train_device(x)
                    k += 1
            elif _relu_kind(m) is not None and isinstance(nxt, LayerNormChannels):
                x = it.LayerNormCh.apply(x, nxt.gamma, nxt.beta, nxt.eps, _relu_kind(m))
                k += 2
            elif isinstance(m, LayerNormChannels):
                x = it.LayerNormCh.apply(x, m.gamma, m.beta, m.eps, None)
                k += 1
            else:
                x = m(x)
                k += 1
        if residual is not None:
            return x, x_fork, False
        return (x, x_fork) if fork else x

    def first_conv_on_device(self, x) -> bool:
        """True when forward(x, in_mul=mask) may be used: the first module is a convolution the HIP kernel serves"""
        return len(self.nn) > 0 and _conv_hip_ok(self.nn[0], x)


class ConvNet(_ConvStack):
    """The reference's ``ConvNet`` (networks.py:247-403).  Spatial path (``len(in_dims) > 1``): see ``__init__``; on 2-D inputs it
    runs on the device passes ``ConvNet2D`` uses.  Vector path (``len(in_dims) == 1``): ``self.nn`` =
    ``Linear(c_in, h0)``, one block per entry of ``c_hidden`` (block i maps ``c_hidden[i-1]`` (``h0`` for i = 0)
    to ``c_hidden[i]``: ``Sequential(f, Linear)`` or ``GatedMLP``, optionally followed by ``LayerNormVector``),
    ``Linear(c_hidden[-1], c_out)`` -- note: no activation in front of the final Linear.  Same constructor
    signature and state-dict keys (``nn.<i>...``) as the reference; convolution arguments are accepted and
    ignored as there."""

    def __init__(self, in_dims, c_hidden: List[int], c_out: int = -1, nonlinearity=nn.ReLU(), kernel_size: int = 3,
                 stride: int = 1, dilation: int = 1, padding: Optional[int] = None, normalize_layers: bool = True,
                 gating: bool = True):
        super().__init__()
        try:
            in_dims = [int(d) for d in in_dims]
        except TypeError:
            raise ValueError("in_dims must be an iterable like [C, H, W] or [C] for vector")
        if padding is None:
            padding = kernel_size // 2
        if not c_hidden or any(h <= 0 for h in c_hidden):
            raise AssertionError("c_hidden must be non-empty list of positive ints")
        c_in = in_dims[0]
        c_out = c_out if c_out > 0 else c_in
        self.c_hidden = [int(h) for h in c_hidden]
        self.gating, self.normalize_layers, self.f = bool(gating), bool(normalize_layers), nonlinearity
        if len(in_dims) != 1:
            # spatial path (networks.py:312-371): Conv, one [GatedConvND | Conv, nonlinearity, (LayerNormChannelsND)] per entry of
            # c_hidden, Conv -- the module tree and state-dict keys of the reference; for 2-D inputs the device passes of
            # ConvNet2D (shared below: _ConvStack) serve it
            rank = max(1, len(in_dims) - 1)
            conv_map = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}
            if rank not in conv_map:
                raise ValueError(f"Unsupported input rank {rank}")
            Conv = conv_map[rank]
            ckw = dict(kernel_size=kernel_size, padding=padding, stride=stride, dilation=dilation)
            mods = [Conv(c_in, self.c_hidden[0], **ckw)]
            width = self.c_hidden[0]
            for h in self.c_hidden:
                if gating:
                    mods += [GatedConvND(width, h, nonlinearity=nonlinearity, input_rank=rank, **ckw), nonlinearity]
                else:
                    mods += [Conv(width, h, **ckw), nonlinearity]
                if normalize_layers:
                    mods.append(LayerNormChannelsND(h, num_spatial_dims=rank))
                width = h
            mods.append(Conv(width, c_out, **ckw))
            self.nn = nn.Sequential(*mods)
            self.is_vector = False
            self._spatial_rank = rank
            return
        mods = [nn.Linear(c_in, self.c_hidden[0])]
        width = self.c_hidden[0]
        for h in self.c_hidden:
            mods.append(GatedMLP(width, h, nonlinearity=nonlinearity) if gating else nn.Sequential(nonlinearity, nn.Linear(width, h)))
            if normalize_layers:
                mods.append(LayerNormVector(h))
            width = h
        mods.append(nn.Linear(width, c_out))
        self.nn = nn.Sequential(*mods)
        self.is_vector = True
        self._vector_in_features = c_in

    def forward(self, x, context=None, in_mul=None, residual=None):
        if not self.is_vector:
            # (L, B, C) / (1, B, C) inputs are accepted as (B, C, L) (networks.py:390-402)
            if x.dim() == 3:
                cin = self.nn[0].in_channels
                if (x.shape[2] == cin and x.shape[0] != x.shape[1]) or (x.shape[0] == 1 and x.shape[2] == cin):
                    x = x.permute(1, 2, 0).contiguous()
            if self._spatial_rank == 2 and x.dim() == 4:
                return _ConvStack.forward(self, x, context, in_mul, residual)
            assert in_mul is None and residual is None
            return self.nn(x)
        x = _as_batch_features(x)
        if x.dim() != 2:
            x = x.view(x.shape[0], -1)
        return self.nn(x)

    # ---- view as a plain MLP, for the fused device path (piece-wise linear configuration only) ----
    def is_plain_mlp(self) -> bool:
        return not self.gating and not self.normalize_layers

    def mlp_view(self):
        """(first Linear, hidden Linears, (W_out, b_out) in fp64, hidden widths).  The last block's Linear and the
        final Linear have no activation between them and are folded into one output map."""
        assert self.is_plain_mlp()
        first = self.nn[0]
        blocks = [self.nn[i][1] for i in range(1, len(self.nn) - 1)]
        final = self.nn[-1]
        Wf, bf = final.weight.detach().double(), final.bias.detach().double()
        Wk, bk = blocks[-1].weight.detach().double(), blocks[-1].bias.detach().double()
        widths = [self.c_hidden[0]] + self.c_hidden[:-1]
        return first, blocks[:-1], (Wf @ Wk, Wf @ bk + bf), widths

    def block_view(self):
        """(first Linear, blocks, final Linear) of the general vector configuration; blocks = one dict per entry of
        ``c_hidden``: ``lin`` (ungated: the Linear behind the activation) or ``l1`` / ``l2`` / ``proj`` (GatedMLP), ``ln``
        (the block's nn.LayerNorm or None), ``w_in`` / ``w_out`` widths"""
        mods = list(self.nn)
        first, final = mods[0], mods[-1]
        blocks, i, width = [], 1, self.c_hidden[0]
        for h in self.c_hidden:
            m = mods[i]
            i += 1
            blk = dict(w_in=width, w_out=h, ln=None)
            if isinstance(m, GatedMLP):
                blk.update(l1=m.net1[1], l2=m.net1[3], proj=m.proj)
            else:
                blk.update(lin=m[1])
            if self.normalize_layers:
                blk["ln"] = mods[i].layernorm
                i += 1
            blocks.append(blk)
            width = h
        assert i == len(mods) - 1
        return first, blocks, final


# ---- CNN conditioners for image-shaped in_dims (networks.py:40-122, 405-510) --------------------------
def _relu_kind(m):
    """(USF_ACT_LEAKY_RELU, slope) for nn.ReLU / nn.LeakyReLU modules (slope 0 = ReLU), else None"""
    from . import _ext
    if isinstance(m, nn.ReLU):
        return (_ext.ACT_LEAKY_RELU, 0.0)
    if isinstance(m, nn.LeakyReLU):
        return (_ext.ACT_LEAKY_RELU, float(m.negative_slope))
    return None


def _conv_same_shaped(conv, x) -> bool:
    """an nn.Conv2d call the device kernels can express: stride 1, "same" zero padding, kernel 1 or 3, fp32 on a ROCm
    device, nothing to differentiate"""
    if not (isinstance(conv, nn.Conv2d) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
        return False
    k = conv.kernel_size
    if k[0] != k[1] or k[0] not in (1, 3) or conv.stride != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1 \
            or conv.padding_mode != "zeros" or x.shape[1] != conv.in_channels:
        return False
    pad = conv.padding
    return pad == "same" or (not isinstance(pad, str) and tuple(pad) == (k[0] // 2, k[0] // 2))


def _conv_hip_ok(conv, x) -> bool:
    """this nn.Conv2d call is served by usf_conv2d_same_f32: `_conv_same_shaped`, and at least two samples per LDS group
    (below that torch's convolution is faster)"""
    if not _conv_same_shaped(conv, x):
        return False
    from . import _ext
    return _ext.load().usf_conv2d_same_fits(conv.in_channels, conv.out_channels, x.shape[2], x.shape[3], conv.kernel_size[0]) >= 2


def _pointwise_hip(conv, x, in_act=None, gate_x=None, post=None):
    """a 1 x 1 nn.Conv2d with few input channels on usf_pointwise_conv_f32 (vector ALUs, exact fp32; HBM-bound work that the
    matrix-core kernel serves at a quarter of the HBM rate), optionally with GatedConv's gate; None if the shape is not served.
    post = ((act id, slope), LayerNormChannels): the nonlinearity and layer norm behind a GatedConv joined to the same pass
    (the caller has checked that the shape allows it)"""
    from . import _ext
    if conv.kernel_size != (1, 1) or not _conv_same_shaped(conv, x) or \
            not _ext.pointwise_conv_supported(conv.in_channels, conv.out_channels, gate_x is not None):
        return None
    key = (conv.weight.data_ptr(), conv.weight._version, None if conv.bias is None else conv.bias._version, str(x.device))
    cache = getattr(conv, "_usf_pointwise", None)
    if cache is None or cache[0] != key:
        w = conv.weight.detach().to(device=x.device, dtype=torch.float32).reshape(conv.out_channels, conv.in_channels).contiguous()
        b = None if conv.bias is None else conv.bias.detach().to(device=x.device, dtype=torch.float32).contiguous()
        cache = conv._usf_pointwise = (key, w, b)
    ia, isl = in_act if in_act is not None else (_ext.ACT_NONE, 0.0)
    oa, osl, ln = _ext.ACT_NONE, 0.0, None
    if post is not None:
        (oa, osl), lnm = post
        ln = (lnm.gamma.detach().reshape(-1).contiguous(), lnm.beta.detach().reshape(-1).contiguous(), lnm.eps)
    if gate_x is not None and config.gated_tail and conv.out_channels == 2 * conv.in_channels and x.dim() == 4 \
            and 0 < x.shape[0] * x.shape[2] * x.shape[3] <= config.gated_tail_infer_max_pixels and _ext.gated_tail_supported(conv.in_channels):
        # few pixels (the reference evaluates in chunks of 100 rows, hyperopt.py:273-278): eight lanes per pixel instead of one
        # thread walking all 2 C dot products -- usf_gated_tail_f32, the training path's forward (live MNIST configuration, 100
        # rows: 19 -> 7 us per launch, 45 launches per log_prob)
        return _ext.gated_tail(x.contiguous(), gate_x.contiguous(), cache[1], cache[2], ia, isl, oa, osl, ln)
    return _ext.pointwise_conv(x.contiguous(), cache[1], cache[2], in_act=ia, in_slope=isl, out_act=oa, out_slope=osl,
                               gate_x=None if gate_x is None else gate_x.contiguous(), ln=ln)


def _conv_hip(conv, x, in_act=None, in_mul=None, out_act=None, residual=None):
    """usf_conv2d_same_f32 for an nn.Conv2d (weight planes cached per parameter version); in_act / out_act: (id, slope);
    residual = (res_x, one_minus_mask, sign): MaskedCoupling's residual joined to the convolution's output stream
    (usf_conv2d_same_res_f32) where the shape allows -- the result is then the coupling's output, flagged by the caller's
    second return value"""
    from . import _ext
    key = (conv.weight.data_ptr(), conv.weight._version, str(x.device))
    cache = getattr(conv, "_usf_planes", None)
    if cache is None or cache[0] != key:
        cache = (key, _ext.conv2d_weight_planes(conv.weight.detach().to(x.device)))
        conv._usf_planes = cache
    ia, isl = in_act if in_act is not None else (_ext.ACT_NONE, 0.0)
    oa, osl = out_act if out_act is not None else (_ext.ACT_NONE, 0.0)
    bias = None if conv.bias is None else conv.bias.detach().to(torch.float32).contiguous()
    if residual is not None and out_act is None and config.conv_res:
        rx, om, sign = residual
        y = _ext.conv2d_same_res(x.contiguous(), cache[1], conv.out_channels, conv.kernel_size[0], rx.contiguous(), om, sign,
                                 bias=bias, in_mul=in_mul, in_act=ia, in_slope=isl)
        if y is not None:
            return y, True
    out = _ext.conv2d_same(x.contiguous(), cache[1], conv.out_channels, conv.kernel_size[0], bias=bias, in_mul=in_mul,
                           in_act=ia, in_slope=isl, out_act=oa, out_slope=osl)
    return (out, False) if residual is not None else out


class LayerNormChannels(nn.Module):
    """layer norm across the channel axis of [B, C, H, W] (networks.py:40-58)"""

    def __init__(self, c_in, eps=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(1, c_in, 1, 1))
        self.beta = nn.Parameter(torch.zeros(1, c_in, 1, 1))
        self.eps = eps

    def _hip_ok(self, x) -> bool:
        return (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3 and x.shape[1] <= 64
                and not (torch.is_grad_enabled() and (x.requires_grad or self.gamma.requires_grad or self.beta.requires_grad)))

    def forward(self, x, pre_act=None):
        """pre_act: None or (USF_ACT id, slope) of a nonlinearity to apply to x first (ConvNet2D folds the (Leaky)ReLU in
        front of this layer into the same device pass)"""
        if self._hip_ok(x):
            from . import _ext
            act, slope = pre_act if pre_act is not None else (_ext.ACT_NONE, 0.0)
            return _ext.layernorm_channels(x.contiguous(), self.gamma.detach().reshape(-1).contiguous(),
                                           self.beta.detach().reshape(-1).contiguous(), self.eps, act, slope)
        assert pre_act is None
        mean = x.mean(dim=1, keepdim=True)
        var = x.var(dim=1, unbiased=False, keepdim=True)
        return (x - mean) / torch.sqrt(var + self.eps) * self.gamma + self.beta


class GatedConv(nn.Module):
    """two-layer convolutional block with a sigmoid input gate: x + val * sigmoid(gate) (networks.py:61-122)"""

    def __init__(self, c_in, c_hidden, kernel_size=3, padding=1, stride=1, nonlinearity=nn.ReLU(), dilation=1):
        super().__init__()
        assert stride == 1, "Stride > 1 cannot be used to skip connection."
        self.net = nn.Sequential(
            nonlinearity,
            nn.Conv2d(c_in, c_hidden, kernel_size=kernel_size, padding=padding, stride=stride, dilation=dilation),
            nonlinearity,
            nn.Conv2d(c_hidden, 2 * c_in, kernel_size=1, padding=padding, stride=stride, dilation=dilation),
        )

    def can_join_layernorm(self, x, ln) -> bool:
        """True when forward(x, post=(act, ln)) computes nonlinearity + LayerNormChannels in the gated convolution's pass:
        the second convolution is 1 x 1 on usf_pointwise_conv_f32 with c_hidden == c_in <= 32 channels"""
        n = self.net
        C = x.shape[1]
        if not (x.is_cuda and x.dim() == 4 and isinstance(ln, LayerNormChannels) and ln._hip_ok(x) and ln.gamma.numel() == C
                and config.pointwise and _relu_kind(n[0]) is not None and _relu_kind(n[2]) is not None):
            return False
        from . import _ext
        c2 = n[3]
        return (isinstance(c2, nn.Conv2d) and c2.kernel_size == (1, 1) and c2.in_channels == C and c2.out_channels == 2 * C
                and C <= 32 and _ext.pointwise_conv_supported(C, 2 * C, True) and _conv_hip_ok(n[1], x)
                and n[1].out_channels == C)

    def train_shape_ok(self, B: int, C: int, H: int, W: int) -> bool:
        """every piece of this block has a device backward at [B, C, H, W] (image_training.py)"""
        from . import image_training as it
        n = self.net
        return (_relu_kind(n[0]) is not None and _relu_kind(n[2]) is not None and isinstance(n[1], nn.Conv2d)
                and isinstance(n[3], nn.Conv2d) and n[1].in_channels == C and n[3].out_channels == 2 * C
                and n[3].in_channels == n[1].out_channels and it.conv_shape_ok(n[1], B, H, W)
                and (it.pointwise_shape_ok(n[3], B, H, W) or it.conv_shape_ok(n[3], B, H, W)))

    def forward_train_device(self, x, post=None):
        """the block as differentiable device passes: 3 x 3 convolution, then -- few pixels (image_training.gated_tail_ok) -- ONE
        pass for the 1 x 1 convolution, the gate, the skip connection and ``post`` = ((act id, slope) | None, LayerNormChannels)
        behind the block; else 1 x 1 convolution and gate as two passes (post must be None)"""
        from . import image_training as it
        n = self.net
        a0, a2 = _relu_kind(n[0]), _relu_kind(n[2])
        # (x forks into the convolution and the skip connection: one node returns both, its backward writes the summed gradient)
        h, x = it.ConvSameFork.apply(x, n[1].weight, n[1].bias, None, a0)
        if post is not None or it.gated_tail_ok(n[3], x):
            pa, lnm = post if post is not None else (None, None)
            return it.GatedTail.apply(h, x, n[3].weight, n[3].bias, None if lnm is None else lnm.gamma,
                                      None if lnm is None else lnm.beta, a2, pa, 0.0 if lnm is None else lnm.eps)
        if it.pointwise_shape_ok(n[3], x.shape[0], x.shape[2], x.shape[3]):
            vg = it.Pointwise.apply(h, n[3].weight, n[3].bias, a2)
        else:
            vg = it.ConvSame.apply(h, n[3].weight, n[3].bias, None, a2, None)
        return it.GatedResidual.apply(x, vg)

    def forward(self, x, post=None):
        """post = ((act id, slope), LayerNormChannels) only after ``can_join_layernorm`` said yes"""
        n = self.net
        a0, a2 = _relu_kind(n[0]), _relu_kind(n[2])
        if post is not None:
            h = _conv_hip(n[1], x, in_act=a0)
            out = _pointwise_hip(n[3], h, in_act=a2, gate_x=x, post=post)
            assert out is not None
            return out
        if a0 is not None and a2 is not None and _conv_hip_ok(n[1], x):
            # device form: the two nonlinearities ride in the convolutions' staging passes, the gate in one more pass
            from . import _ext
            h = _conv_hip(n[1], x, in_act=a0)
            C = x.shape[1]
            if n[3].out_channels == 2 * C and config.pointwise:
                out = _pointwise_hip(n[3], h, in_act=a2, gate_x=x)      # second convolution + gate: one pass on the vector ALUs
                if out is not None:
                    return out
            if _conv_hip_ok(n[3], h) and n[3].out_channels == 2 * C:
                cout_g = 32 * ((C + 15) // 16)
                if cout_g <= 64 and _ext.load().usf_conv2d_same_fits(h.shape[1], cout_g, h.shape[2], h.shape[3],
                                                                        n[3].kernel_size[0]) >= 2:
                    # the second convolution and the gate in ONE launch: its (value, gate) rows are packed pairwise and
                    # the epilogue writes x + value * sigmoid(gate) -- the [B, 2C, H, W] tensor never exists
                    conv = n[3]
                    key = (conv.weight.data_ptr(), conv.weight._version, str(x.device), "gate")
                    cache = getattr(conv, "_usf_gate_planes", None)
                    if cache is None or cache[0] != key:
                        w = conv.weight.detach().to(x.device)
                        rows = _ext.conv2d_gate_row_order(C, x.device)
                        ok = _ext.conv2d_gate_row_order(C, x.device, valid=True)
                        bias = None if conv.bias is None else (conv.bias.detach().to(torch.float32)[rows] * ok).contiguous()
                        cache = conv._usf_gate_planes = (key, _ext.conv2d_weight_planes(w, gate_channels=C), bias)
                    return _ext.conv2d_same(h, cache[1], cout_g, conv.kernel_size[0], bias=cache[2], in_act=a2[0],
                                            in_slope=a2[1], gate_x=x.contiguous())
                vg = _conv_hip(n[3], h, in_act=a2)
                if vg.shape[1] == 2 * C and vg.shape[2:] == x.shape[2:]:
                    return _ext.gated_residual(x.contiguous(), vg)
        vg = self.net(x)
        if (x.is_cuda and x.dtype == torch.float32 and vg.dtype == torch.float32 and vg.shape[1] == 2 * x.shape[1]
                and vg.shape[2:] == x.shape[2:] and not (torch.is_grad_enabled() and (x.requires_grad or vg.requires_grad))):
            from . import _ext
            return _ext.gated_residual(x.contiguous(), vg.contiguous())        # one pass instead of chunk / sigmoid / mul / add
        val, gate = vg.chunk(2, dim=1)
        ret = x + val * torch.sigmoid(gate)
        assert ret.shape == x.shape, f"Shape mismatch: {ret.shape} != {x.shape}"
        return ret


class LayerNormChannelsND(LayerNormChannels):
    """channel-wise layer norm of (batch, channel, *spatial) tensors (networks.py:122-141): gamma / beta shaped
    (1, C, 1, ..., 1) with ``num_spatial_dims`` trailing ones; the same arithmetic (and device pass) as LayerNormChannels"""

    def __init__(self, c_in, num_spatial_dims: int = 2, eps=1e-5):
        nn.Module.__init__(self)
        shape = (1, c_in) + (1,) * num_spatial_dims
        self.gamma = nn.Parameter(torch.ones(*shape))
        self.beta = nn.Parameter(torch.zeros(*shape))
        self.eps = eps


class GatedConvND(GatedConv):
    """gated residual block for 1 / 2 / 3-D convolutions (networks.py:144-203): ``x' + val * sigmoid(gate)``,
    ``[val, gate] = Conv1x1(f(Conv(f(x))))``, ``x'`` = x or a 1 x 1 projection when the channel counts differ.  (Unlike the
    2-D ``GatedConv`` the pointwise convolution has padding 0 and 2 * c_out outputs.)  With 2-D inputs and no projection it is
    GatedConv's module sequence and takes its device passes."""

    def __init__(self, c_in, c_out, kernel_size=3, padding=1, stride=1, dilation=1, nonlinearity=nn.ReLU(), input_rank: int = 2):
        nn.Module.__init__(self)
        assert stride == 1, "Stride > 1 cannot be used to skip connection."
        conv_map = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}
        if input_rank not in conv_map:
            raise ValueError(f"Unsupported input rank {input_rank}")
        Conv = conv_map[input_rank]
        self.net = nn.Sequential(
            nonlinearity,
            Conv(c_in, c_out, kernel_size=kernel_size, padding=padding, stride=stride, dilation=dilation),
            nonlinearity,
            Conv(c_out, 2 * c_out, kernel_size=1, padding=0, stride=1),
        )
        self.proj = Conv(c_in, c_out, kernel_size=1, padding=0) if c_in != c_out else None
        self._plain2d = input_rank == 2 and self.proj is None

    def can_join_layernorm(self, x, ln) -> bool:
        return self._plain2d and super().can_join_layernorm(x, ln)

    def train_shape_ok(self, B, C, H, W) -> bool:
        return self._plain2d and super().train_shape_ok(B, C, H, W)

    def forward(self, x, post=None):
        if self._plain2d and x.dim() == 4:
            return super().forward(x, post)
        assert post is None
        val, gate = self.net(x).chunk(2, dim=1)
        skip = x if self.proj is None else self.proj(x)
        return skip + val * torch.sigmoid(gate)


class ConvNet2D(_ConvStack):
    """the CNN conditioner of the reference's image configs (networks.py:405-510): Conv2d(c_in, c_hidden), then
    ``num_layers`` x [GatedConv | Conv2d, nonlinearity, (LayerNormChannels)], then Conv2d(c_hidden, c_out).  Same
    module tree (``nn.{i}``) and state-dict keys; runs as torch ops (MIOpen convolutions on a ROCm device)."""

    def __init__(self, c_in: int, c_hidden: int = 3, c_out: int = -1, num_layers: int = 3, nonlinearity=nn.ReLU(),
                 kernel_size: int = 3, stride: int = 1, dilation: int = 1, padding=0, normalize_layers: bool = True,
                 gating: bool = True):
        super().__init__()
        if padding is None:
            padding = kernel_size // 2
        self.nonlinearity = nonlinearity
        c_out = c_out if c_out > 0 else c_in
        conv = lambda a, b: nn.Conv2d(a, b, kernel_size=kernel_size, padding=padding, stride=stride, dilation=dilation)
        layers = [conv(c_in, c_hidden)]
        for _ in range(num_layers):
            if gating:
                layers += [GatedConv(c_hidden, c_hidden, kernel_size=kernel_size, padding=padding, stride=stride,
                                     dilation=dilation), nonlinearity]
            else:
                layers += [conv(c_hidden, c_hidden), nonlinearity]
            if normalize_layers:
                layers += [LayerNormChannels(c_hidden)]
        layers += [conv(c_hidden, c_out)]
        self.nn = nn.Sequential(*layers)

