"""Generate golden vectors by running the REAL reference (this container only).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Imports /root/reference through ``ref_shim`` (plumbing-only pyro stand-in), builds reference
``USFlow`` models, and stores inputs + the reference's outputs (fp32 run and fp64 run) as
small ``.npz`` fixtures.  Fixtures are data only: parameters, inputs, expected outputs.
Nothing of the reference's source travels.

Two families of cases:
* ``init``  -- parameters exactly as the reference constructors draw them under
               ``torch.manual_seed(seed)`` (small K so the default init stays finite);
* ``synth`` -- parameters from ``oracle.usflows_oracle.synth_state_dict`` (the documented
               well-conditioned generator) loaded into the reference model with
               ``load_state_dict(strict=True)`` -- this also pins the state-dict key layout.
For D=784,K=32 (BASELINE cfg2) the 55 M-parameter state dict is NOT stored: it is regenerated
from (spec, seed) by the same generator on the consumer side; only x[:64] and outputs are kept.
"""
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_shim  # noqa: E402
from oracle import usflows_oracle as orc  # noqa: E402

flows, transforms, networks, distributions = ref_shim.install()


def make_base(spec, dtype=torch.float32):
    D = spec.dim
    if spec.base == "laplace":
        loc = spec.base_loc if spec.base_loc is not None else torch.zeros(D)
        sc = spec.base_scale if spec.base_scale is not None else torch.ones(D)
        return torch.distributions.Laplace(loc.to(dtype), sc.to(dtype))
    if spec.base == "normal":
        loc = spec.base_loc if spec.base_loc is not None else torch.zeros(D)
        sc = spec.base_scale if spec.base_scale is not None else torch.ones(D)
        return torch.distributions.Normal(loc.to(dtype), sc.to(dtype))
    if spec.base == "radial":
        if spec.radial_norm == "gammamm":
            k = int(spec.extra.get("gammamm_k", 4))
            nd = distributions.GammaMM(torch.linspace(2.0, 6.0, k), torch.ones(k), torch.ones(k) / k)
        else:
            nd = distributions.LogNormal(torch.tensor([spec.radial_norm_loc]),
                                         torch.tensor([spec.radial_norm_scale]))
        loc = spec.base_loc if spec.base_loc is not None else torch.zeros(D)
        rd = distributions.RadialDistribution(loc.clone(), nd, float(spec.radial_p))
        return rd.to(dtype)
    raise ValueError(spec.base)


def build_reference(spec, seed):
    torch.manual_seed(seed)
    act = torch.nn.LeakyReLU(spec.negative_slope) if spec.negative_slope != 0 else torch.nn.ReLU()
    if spec.conditioner == "ConditionalDenseNN":
        cls = networks.ConditionalDenseNN
        args = dict(input_dim=spec.dim, context_dim=1, hidden_dims=list(spec.hidden_dims),
                    out_dim=spec.dim, nonlinearity=act)
    elif spec.conditioner == "ConvNet":
        cls = networks.ConvNet
        args = dict(in_dims=[spec.dim], c_hidden=list(spec.hidden_dims), nonlinearity=act,
                    normalize_layers=bool(spec.extra.get("normalize_layers", False)), gating=bool(spec.extra.get("gating", False)))
    else:
        from pyro.nn import DenseNN
        cls = DenseNN
        args = dict(input_dim=spec.dim, hidden_dims=list(spec.hidden_dims), param_dims=[spec.dim],
                    nonlinearity=act)
    prior = torch.distributions.Uniform(1e-20, 0.01) if spec.soft_training else None
    flow = flows.USFlow(make_base(spec), [spec.dim], spec.coupling_blocks, cls, args,
                        soft_training=spec.soft_training, training_noise_prior=prior,
                        affine_conjugation=spec.affine_conjugation,
                        lu_transform=spec.lu_transform, householder=spec.householder)
    return flow


def to_double(flow, spec):
    flow = flow.double()
    for l in flow.layers:
        if isinstance(l, transforms.MaskedCoupling):
            l.mask = l.mask.double()
    b = make_base(spec, torch.float64)
    if spec.base == "radial" and spec.radial_norm == "gammamm":
        # (this runs under default dtype float64: the GammaMM constructor would draw its parameters afresh in fp64;
        # the fp64 run must see the SAME parameters as the fp32 run and the stored state dict)
        b.load_state_dict({k: v.double() for k, v in flow.base_distribution.state_dict().items()})
    if len(getattr(b, "batch_shape", ())) > 0:
        b = distributions.Independent(b, 1)
    flow.base_distribution = b
    return flow


def spec_to_json(spec):
    d = dict(dim=spec.dim, coupling_blocks=spec.coupling_blocks, hidden_dims=list(spec.hidden_dims),
             lu_transform=spec.lu_transform, householder=spec.householder,
             affine_conjugation=spec.affine_conjugation, negative_slope=spec.negative_slope,
             conditioner=spec.conditioner, base=spec.base, radial_p=("inf" if spec.radial_p == math.inf else spec.radial_p),
             radial_norm=spec.radial_norm, radial_norm_loc=spec.radial_norm_loc,
             radial_norm_scale=spec.radial_norm_scale, soft_training=spec.soft_training, extra=dict(spec.extra))
    return json.dumps(d)


ONLY = None      # --only <substring>: regenerate just the matching cases


def run_case(name, spec, family, seed, n=48, store_sd=True, ctx=False, alpha=0.1, x_scale=1.0):
    if ONLY is not None and ONLY not in name:
        return
    flow = build_reference(spec, seed)
    if family == "synth":
        sd = orc.synth_state_dict(spec, seed=seed, alpha=alpha)
        res = flow.load_state_dict(sd, strict=False)   # strict except for base-distribution params
        assert not res.unexpected_keys, res.unexpected_keys
        assert all(k.startswith("base_distribution.") for k in res.missing_keys), res.missing_keys
    sd = {k: v.detach().clone() for k, v in flow.state_dict().items()} if store_sd else {}
    g = torch.Generator().manual_seed(1000 + seed)
    x = torch.rand(n, spec.dim, generator=g) * x_scale
    zin = torch.distributions.Laplace(0.0, 1.0).icdf(torch.rand(n, spec.dim, generator=g) * 0.998 + 0.001)
    context = None
    if ctx:
        context = torch.rand(n, 1, generator=g) * 2
    out = {}
    with torch.no_grad():
        out["log_prob32"] = flow.log_prob(x, context=context) if ctx else flow.log_prob(x)
        if ctx:
            y = x
            for l in reversed(flow.layers):
                y = l.backward(y, context=context)
            out["backward32"] = y
            y = zin
            for l in flow.layers:
                y = l.forward(y, context=context)
            out["forward32"] = y
        else:
            out["backward32"] = flow.backward(x)
            out["forward32"] = flow._forward(zin)
        # the reference builds torch.eye()/zeros() at default dtype inside matrix()/bias()
        # (transforms.py:1459,1466,1473) -> an fp64 run needs the default dtype switched too
        torch.set_default_dtype(torch.float64)
        flow64 = to_double(flow, spec)
        x64, z64 = x.double(), zin.double()
        c64 = context.double() if ctx else None
        out["log_prob64"] = flow64.log_prob(x64, context=c64) if ctx else flow64.log_prob(x64)
        if ctx:
            y = x64
            for l in reversed(flow64.layers):
                y = l.backward(y, context=c64)
            out["backward64"] = y
            y = z64
            for l in flow64.layers:
                y = l.forward(y, context=c64)
            out["forward64"] = y
        else:
            out["backward64"] = flow64.backward(x64)
            out["forward64"] = flow64._forward(z64)
        ladj = 0.0
        for l in flow64.layers:
            ladj = ladj + l.log_abs_det_jacobian(None, None)
        out["total_ladj64"] = torch.as_tensor(ladj, dtype=torch.float64)
        torch.set_default_dtype(torch.float32)
    arrays = {"x": x.numpy(), "zin": zin.numpy()}
    if ctx:
        arrays["context"] = context.numpy()
    for k, v in out.items():
        arrays[k] = v.detach().numpy()
    if store_sd:
        for k, v in sd.items():
            arrays["sd/" + k] = v.numpy()
    if spec.base_loc is not None:
        arrays["base_loc"] = spec.base_loc.numpy()
    if spec.base_scale is not None:
        arrays["base_scale"] = spec.base_scale.numpy()
    arrays["spec"] = np.array(spec_to_json(spec))
    arrays["family"] = np.array(family)
    arrays["seed"] = np.array(seed)
    arrays["alpha"] = np.array(alpha)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    rel = (out["log_prob32"].double() - out["log_prob64"]).abs() / out["log_prob64"].abs()
    print(f"{name:42s} logp[0]={out['log_prob64'][0].item():+.6e} ref32-vs-64 max rel {rel.max().item():.2e} "
          f"max|z|={out['backward64'].abs().max().item():.3g}  {os.path.getsize(path)/1024:.0f} KB")


def main():
    S = orc.FlowSpec
    # --- reference-initialised ("init") small cases ------------------------------------
    run_case("init_d2_k4_hh0_laplace", S(2, 4, [32, 32], householder=0, negative_slope=0.0), "init", 1)
    run_case("init_d7_k2_hh1_conj_laplace", S(7, 2, [16, 16], householder=1, affine_conjugation=True), "init", 2)
    run_case("init_d16_k2_hh2_normal", S(16, 2, [24], householder=2, base="normal"), "init", 3)
    run_case("init_d2_k3_densenn_conj", S(2, 3, [32, 32], householder=0, affine_conjugation=True,
                                           conditioner="DenseNN", negative_slope=0.0), "init", 4)
    # --- well-conditioned ("synth") cases ----------------------------------------------
    run_case("synth_d2_k4_hh0_laplace", S(2, 4, [32, 32], householder=0, negative_slope=0.0), "synth", 11)
    run_case("synth_d7_k3_hh0_laplace", S(7, 3, [16, 16], householder=0), "synth", 12)
    run_case("synth_d7_k3_hh1_conj_normal", S(7, 3, [16, 8], householder=1, affine_conjugation=True,
                                               base="normal",
                                               base_loc=torch.linspace(-0.5, 0.5, 7),
                                               base_scale=torch.linspace(0.5, 2.0, 7)), "synth", 13)
    run_case("synth_d16_k4_hh2_conj_laplace", S(16, 4, [32, 32, 32], householder=2, affine_conjugation=True,
                                                base_loc=torch.linspace(-1, 1, 16),
                                                base_scale=torch.linspace(0.7, 1.3, 16)), "synth", 14)
    run_case("synth_d16_k4_hh0_conj_radial1", S(16, 4, [32, 32], householder=0, affine_conjugation=True,
                                                base="radial", radial_p=1.0, radial_norm_loc=2.0,
                                                radial_norm_scale=0.35), "synth", 15)
    run_case("synth_d16_k3_hh1_radial2", S(16, 3, [32], householder=1, base="radial", radial_p=2.0,
                                           radial_norm_loc=1.0, radial_norm_scale=0.5), "synth", 16)
    run_case("synth_d16_k3_hh0_radialinf", S(16, 3, [32], householder=0, base="radial", radial_p=math.inf,
                                             radial_norm_loc=0.5, radial_norm_scale=0.5), "synth", 17)
    # radial base with the live configs' norm distribution: a Gamma mixture (gaussian_mixture.yaml:84-93), p = 1
    run_case("synth_d16_k3_hh0_conj_radial1_gammamm", S(16, 3, [32, 32], householder=0, affine_conjugation=True,
                                                        base="radial", radial_p=1.0, radial_norm="gammamm",
                                                        extra={"gammamm_k": 5}), "synth", 26)
    run_case("synth_d64_k6_hh0_laplace", S(64, 6, [96, 64], householder=0), "synth", 18)
    run_case("synth_d64_k4_hh1_conj_laplace", S(64, 4, [64, 64], householder=1, affine_conjugation=True), "synth", 19)
    run_case("synth_d7_k3_soft_ctx", S(7, 3, [16, 16], householder=0, soft_training=True), "synth", 20, ctx=True)
    run_case("synth_d7_k3_soft_noctx", S(7, 3, [16, 16], householder=0, soft_training=True), "synth", 21)
    run_case("synth_d33_k3_lu2_hh1", S(33, 3, [40, 24], lu_transform=2, householder=1), "synth", 22)
    run_case("synth_d16_k3_densenn_relu", S(16, 3, [32, 32], householder=0, conditioner="DenseNN",
                                            negative_slope=0.0), "synth", 23)
    # --- vector path of the generic ConvNet conditioner (gating=False, normalize_layers=False) ---
    run_case("synth_d16_k3_convnet_vec", S(16, 3, [32, 24], householder=0, conditioner="ConvNet"), "synth", 24)
    run_case("synth_d33_k2_convnet_vec_conj", S(33, 2, [40], householder=1, affine_conjugation=True,
                                               conditioner="ConvNet", negative_slope=0.0), "synth", 25)
    # --- the same class as the reference constructs it by default: GatedMLP blocks + LayerNormVector (networks.py:206-245) ---
    run_case("synth_d16_k3_convnet_gated_ln", S(16, 3, [32, 24], householder=0, conditioner="ConvNet",
                                                extra={"gating": True, "normalize_layers": True}), "synth", 27)
    run_case("synth_d33_k2_convnet_gated_conj", S(33, 2, [40, 40], householder=1, affine_conjugation=True, conditioner="ConvNet",
                                                  negative_slope=0.0, extra={"gating": True}), "synth", 28)
    run_case("synth_d64_k3_convnet_ln", S(64, 3, [96, 50], householder=0, conditioner="ConvNet",
                                          extra={"normalize_layers": True}), "synth", 29)
    run_case("init_d4_k2_convnet_default", S(4, 2, [32, 32], householder=1, conditioner="ConvNet", negative_slope=0.0,
                                              extra={"gating": True, "normalize_layers": True}), "init", 5)
    # --- BASELINE cfg2 model (D=784, K=32, h=[256,256]); state dict regenerated from seed --
    run_case("synth_d784_k32_cfg2", S(784, 32, [256, 256], householder=0), "synth", 100, n=64, store_sd=False)
    run_case("synth_d784_k4_hh1_conj", S(784, 4, [256, 256], householder=1, affine_conjugation=True),
             "synth", 101, n=32, store_sd=False)
    # --- the two variants SURVEY section 8 wants parity-tested AND reported beside the headline, at the cfg2 depth, and the
    # secondary base of section 8d (Radial p = 1, LogNormal(6, .35): tests/explib/mnist.yaml:79-92) on the flat cfg2 model;
    # bench.py places these rows at the head / middle / tail of its 65536-row batches (--golden <name>)
    run_case("synth_d784_k32_conj", S(784, 32, [256, 256], householder=0, affine_conjugation=True),
             "synth", 103, n=32, store_sd=False)
    run_case("synth_d784_k32_hh1_conj", S(784, 32, [256, 256], householder=1, affine_conjugation=True),
             "synth", 104, n=32, store_sd=False)
    run_case("synth_d784_k32_radial1", S(784, 32, [256, 256], householder=0, base="radial", radial_p=1.0,
                                         radial_norm_loc=6.0, radial_norm_scale=0.35), "synth", 105, n=32, store_sd=False)
    # --- BASELINE cfg4 model (D=3072, K=48, h=[1024,1024]; 1.28 G parameters): 16 rows, outputs only -------------
    # (each pass of the reference re-inverts 49 x 2 triangular 3072 x 3072 factors: ~10 min and ~25 GB here)
    run_case("synth_d3072_k48_cfg4", S(3072, 48, [1024, 1024], householder=0), "synth", 102, n=16, store_sd=False)


if __name__ == "__main__":
    if "--only" in sys.argv:
        ONLY = sys.argv[sys.argv.index("--only") + 1]
    main()
