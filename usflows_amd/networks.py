"""Conditioner networks of the hot path (flat inputs): dense MLPs.

``ConditionalDenseNN`` mirrors the reference's in-repo class (networks.py:681-751: same ctor
arguments, ``layers`` ModuleList ordering [input, context, hidden..., output] and therefore the
same state-dict keys).  ``DenseNN`` stands in for ``pyro.nn.DenseNN`` (pyro-ppl 1.8.6), which the
live vector configs name by dotted path (experiments/synthetic/gaussian_mixture.yaml:66-71);
pyro is not a dependency of this package.  CNN conditioners are out of scope.

The modules themselves are plain torch (used under autograd / on CPU); on the device fast path
``usflows_amd.engine`` reads their parameters and runs the whole MLP inside the fused coupling
kernel -- ``forward`` below is not called there.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
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
