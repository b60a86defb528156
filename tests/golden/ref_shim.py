"""Import harness for the *reference* USFlows package (this container only).

TEST INFRASTRUCTURE -- never imported by the product (``usflows_amd``), by ``bench.py``
or by the ``-m gpu`` tests.  It is used only by ``make_golden.py`` (fixture generator)
and by the optional ``test_oracle_vs_reference`` cross-check, which skips itself when
``/root/reference`` is absent (it is absent on the GPU box).

The reference hard-imports ``pyro`` (flows.py:5, transforms.py:7-14, networks.py:9,
distributions.py:5,16) which is not installed here and cannot be (no network).  The
hot-path arithmetic is all ``torch``; what the reference takes from pyro is class plumbing:

* ``pyro.distributions``          -> ``torch.distributions`` re-export
* ``pyro.distributions.TransformModule`` = ``torch.distributions.Transform`` + ``nn.Module``
* ``pyro.distributions.transforms.Permute`` / ``pyro.infer.SVI`` -> names only (unused)
* ``pyro.nn.DenseNN`` -> the one arithmetic class: a plain MLP.  pyro-ppl 1.8.6
  (poetry.lock:3198) is not vendored under /root/reference, so its published behaviour is
  restated here: ``Linear(in,h0) -> f -> Linear(h0,h1) -> f ... -> Linear(h_last, sum(param_dims))``,
  no activation on the output, output returned as one tensor when ``len(param_dims)==1``.
  Golden vectors therefore use the in-repo ``ConditionalDenseNN`` (networks.py:681-751) as the
  canonical conditioner; DenseNN parity is "unpinned" (SURVEY.md section 8c).

The shim is written to a temp dir at import time; nothing of the reference is copied.
"""
import os
import sys
import tempfile
import textwrap

REFERENCE_ROOT = os.environ.get("USFLOWS_REFERENCE_ROOT", "/root/reference")

_PYRO_INIT = """
from . import distributions, nn, infer
"""

_PYRO_DIST = """
import torch
from torch.distributions import *            # noqa: F401,F403
from torch.distributions import constraints, transforms, Distribution, TransformedDistribution
from torch.distributions import Independent, Laplace, Normal, Uniform, Dirichlet, Categorical
from . import transforms as transforms       # pyro.distributions.transforms (Permute name only)


class TransformedDistribution(TransformedDistribution):
    # pyro's subclass adds clear_cache() (Flow.fit calls it after every optimiser step, flows.py:207): drops the
    # cached (x, y) pairs of the transforms -- plumbing, no arithmetic
    def clear_cache(self):
        for t in self.transforms:
            if getattr(t, "_cache_size", 0) == 1:
                t._cached_x_y = None, None


class TransformModule(torch.distributions.Transform, torch.nn.Module):
    # class plumbing only: pyro's TransformModule is exactly this multiple inheritance
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    def __hash__(self):
        return torch.nn.Module.__hash__(self)
"""

_PYRO_DIST_TRANSFORMS = """
from torch.distributions.transforms import *  # noqa: F401,F403


class Permute:                                 # name only; shadowed at transforms.py:174
    pass
"""

_PYRO_INFER = """
class SVI:                                     # name only; unused on the hot path
    pass
"""

_PYRO_NN = """
import torch


class DenseNN(torch.nn.Module):
    # restatement of pyro-ppl 1.8.6 pyro.nn.DenseNN (see ref_shim.py docstring)
    def __init__(self, input_dim, hidden_dims, param_dims=[1, 1], nonlinearity=torch.nn.ReLU()):
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dims = hidden_dims
        self.param_dims = param_dims
        self.count_params = len(param_dims)
        self.output_multiplier = sum(param_dims)
        ends = torch.cumsum(torch.tensor(param_dims), dim=0)
        starts = torch.cat((torch.zeros(1).type_as(ends), ends[:-1]))
        self.param_slices = [slice(s.item(), e.item()) for s, e in zip(starts, ends)]
        layers = [torch.nn.Linear(input_dim, hidden_dims[0])]
        for i in range(1, len(hidden_dims)):
            layers.append(torch.nn.Linear(hidden_dims[i - 1], hidden_dims[i]))
        layers.append(torch.nn.Linear(hidden_dims[-1], self.output_multiplier))
        self.layers = torch.nn.ModuleList(layers)
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
        return tuple([h[..., s] for s in self.param_slices])
"""


def install():
    """Put a plumbing-only ``pyro`` and the reference root on sys.path.  Returns the
    reference modules (flows, transforms, networks, distributions)."""
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "usflows")):
        raise FileNotFoundError(f"reference not present at {REFERENCE_ROOT}")
    if "pyro" not in sys.modules:
        d = tempfile.mkdtemp(prefix="pyro_shim_")
        pkg = os.path.join(d, "pyro")
        os.makedirs(os.path.join(pkg, "distributions"))
        files = {
            "__init__.py": _PYRO_INIT,
            "distributions/__init__.py": _PYRO_DIST,
            "distributions/transforms.py": _PYRO_DIST_TRANSFORMS,
            "infer.py": _PYRO_INFER,
            "nn.py": _PYRO_NN,
        }
        for name, body in files.items():
            with open(os.path.join(pkg, name), "w") as f:
                f.write(textwrap.dedent(body))
        sys.path.insert(0, d)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # /root/reference is read-only
    import src.usflows.flows as flows
    import src.usflows.transforms as transforms
    import src.usflows.networks as networks
    import src.usflows.distributions as distributions
    return flows, transforms, networks, distributions
