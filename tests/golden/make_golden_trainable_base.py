"""Golden gradients for flows whose base is one of the reference's TRAINABLE distribution modules -- ``distributions.Laplace`` /
``distributions.Normal`` (src/usflows/distributions.py:199-238: ``loc`` and a softplus-constrained ``scale_unconstrained`` as
nn.Parameters) -- from the REAL reference (this container only):

    python tests/golden/make_golden_trainable_base.py     # writes tests/golden/tbase_<case>.npz

Per case: the state dict (layers + base), inputs, ``log_prob`` in fp32 / fp64 and the fp64 gradient of Flow.fit's loss
``-log_prob(x).mean()`` (flows.py:196-198) w.r.t. EVERY parameter, the base's ``loc`` / ``scale_unconstrained`` included.
Layer parameters: oracle/synth.py's conditioned generator.  Data only."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_golden as mg  # noqa: E402  (imports the reference through ref_shim; main() is not run)
from oracle import usflows_oracle as orc  # noqa: E402

distributions = mg.distributions


def run_case(name, spec, seed, loc, scale, n=48):
    flow = mg.build_reference(spec, seed)                   # (its fixed base is replaced below, before anything is evaluated)
    sd = orc.synth_state_dict(spec, seed=seed, alpha=0.1)
    res = flow.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    cls = distributions.Laplace if spec.base == "laplace" else distributions.Normal
    base = cls(loc.clone(), scale.clone())
    # exactly what Flow.__init__ does with a base distribution (flows.py:90-103)
    flow.base_distribution = base
    assert len(base.batch_shape) == 0
    import pyro.distributions as dist
    flow.transform = dist.TransformedDistribution(base, flow.layers)
    x = torch.rand(n, spec.dim, generator=torch.Generator().manual_seed(1000 + seed))
    with torch.no_grad():
        lp32 = flow.log_prob(x)
    full_sd = {k: v.detach().clone() for k, v in flow.state_dict().items()}
    torch.set_default_dtype(torch.float64)
    try:
        f64 = flow.double()
        for l in f64.layers:
            if isinstance(l, mg.transforms.MaskedCoupling):
                l.mask = l.mask.double()
        for q in f64.parameters():
            q.grad = None
        lp = f64.log_prob(x.double())
        (-lp.mean()).backward()
        grads = {k: q.grad.detach().clone() for k, q in f64.named_parameters() if q.grad is not None}
    finally:
        torch.set_default_dtype(torch.float32)
    assert "base_distribution.loc" in grads and "base_distribution.scale_unconstrained" in grads
    arrays = {"x": x.numpy(), "log_prob32": lp32.numpy(), "log_prob64": lp.detach().numpy(), "loss64": np.array(float(-lp.mean()))}
    arrays.update({"sd/" + k: v.float().numpy() for k, v in full_sd.items()})
    arrays.update({"g/" + k: v.numpy() for k, v in grads.items()})
    arrays["spec"] = np.array(mg.spec_to_json(spec))
    arrays["seed"] = np.array(seed)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    rel = (lp32.double() - lp.detach()).abs() / lp.detach().abs()
    print(f"{name:40s} logp[0]={lp[0].item():+.6e} ref32-vs-64 {rel.max().item():.2e} {len(grads)} grads {os.path.getsize(path) / 1024:.0f} KB")


def main():
    S = orc.FlowSpec
    run_case("tbase_d7_k3_hh1_conj_normal", S(7, 3, [16, 8], householder=1, affine_conjugation=True, base="normal",
                                              extra={"trainable_base": True}), 71,
             torch.linspace(-0.5, 0.5, 7), torch.linspace(0.5, 2.0, 7))
    run_case("tbase_d16_k3_hh0_laplace", S(16, 3, [32, 32], householder=0, base="laplace", extra={"trainable_base": True}), 72,
             torch.linspace(-1, 1, 16), torch.linspace(0.7, 1.3, 16))
    run_case("tbase_d16_k3_hh0_normal_scalar_scale", S(16, 3, [32], householder=0, base="normal",
                                                       extra={"trainable_base": "scalar_scale"}), 73,
             torch.linspace(-1, 1, 16), torch.tensor(1.3))
    run_case("tbase_d160_k2_hh0_laplace", S(160, 2, [96, 64], householder=0, base="laplace", extra={"trainable_base": True}), 74,
             torch.linspace(-1, 1, 160), torch.linspace(0.7, 1.3, 160), n=40)


if __name__ == "__main__":
    main()
