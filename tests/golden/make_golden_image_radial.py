"""Golden vectors of the reference's LIVE image configurations -- image-shaped flow, ``RadialDistribution`` base with an
image-shaped ``loc``, ``prior_scale: 1.0`` -- from the REAL reference (this container only).

    python tests/golden/make_golden_image_radial.py     # writes tests/golden/imageradial_<case>.npz, imageradialfit_<case>.npz

What ``HyperoptExperiment._trial`` trains (experiments/mnist/mnist.yaml:30-92: ``in_dims [16,7,7]``, ``ConvNet2D(c_hidden 32,
num_layers 3, gated, layer norm)``, ``lu_transform 1, householder 0, affine_conjugation true``, ``prior_scale 1.0``, base
``RadialDistribution(zeros[16,7,7], p=1, LogNormal(6, .35))``, batch 32, SophiaG lr 1e-3 weight_decay 0;
experiments/fashion/fashionclasses_veriflow.yaml:55-93: the same flow with 10 blocks over ``GammaMM`` x 20;
experiments/cifar/cifar.yaml: ``in_dims [48,8,8]``).  The layer parameters come from tests/image_synth.py (a pure function
of the seed and the module structure; the default initialisation explodes); the base distribution's own
parameters are what the configuration's constructor calls give (drawn under ``torch.manual_seed`` for GammaMM) and are
stored.  Per case: inputs, ``log_prob`` / ``backward`` / ``_forward`` in fp32 and fp64, the fp64 gradients of ``Flow.fit``'s
loss ``-log_prob(x).mean() - log_prior()`` (flows.py:196-198) w.r.t. EVERY parameter (base included) and the value of
``log_prior()``.  Fit cases: 2 epochs x 96 rows, batch 32, of ``Flow.fit`` with its default optimiser SophiaG at the live
hyper-parameters, fp32 on the CPU: per-epoch losses and every parameter after the 6 steps.  Data only."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_shim  # noqa: E402

flows, transforms, networks, distributions = ref_shim.install()
from image_synth import synth_image_params_  # noqa: E402

# (the configurations also pass ``rescale_hidden: 1``, which the reference's ConvNet2D.__init__ -- networks.py:405-440 -- does
# not accept any more: the YAML is one argument behind the code; dropped here, everything else as written)
LIVE_COND = dict(c_hidden=32, num_layers=3, padding="same", kernel_size=3, normalize_layers=True, gating=True)
LR, NP_SEED, N_ROWS, BATCH, EPOCHS = 1e-3, 5, 96, 32, 2


def make_base(kind, in_dims, p=1.0):
    """the configurations' base_distribution entries, constructor call for constructor call"""
    if kind == "lognormal":         # mnist.yaml:79-92
        nd = distributions.LogNormal(loc=torch.ones([1]) * 6, scale=torch.ones([1]) * .35, device="cpu")
    elif kind == "gammamm":         # fashionclasses_veriflow.yaml:79-93
        nd = distributions.GammaMM(concentration=torch.rand([20]) * 75, rate=torch.rand([20]),
                                   mixture_weights=torch.ones([20]) / 20, device="cpu")
    else:
        raise KeyError(kind)
    return distributions.RadialDistribution(device="cpu", p=float(p), loc=torch.zeros(list(in_dims)), norm_distribution=nd)


def build(spec):
    torch.manual_seed(spec["seed"])
    base = make_base(spec["base"], spec["in_dims"], spec["p"])
    cond = dict(LIVE_COND, c_in=spec["in_dims"][0])
    flow = flows.USFlow(base, list(spec["in_dims"]), spec["coupling_blocks"], networks.ConvNet2D, cond, prior_scale=1.0,
                        lu_transform=1, householder=0, affine_conjugation=True, nonlinearity=torch.nn.ReLU())
    synth_image_params_(flow, spec["seed"])
    if spec.get("loc_noise"):
        # a non-zero loc so that its gradient and the subtraction are exercised (the configurations start it at zero and
        # train it: RadialDistribution.loc is an nn.Parameter, distributions.py:369)
        with torch.no_grad():
            flow.base_distribution.loc.copy_(spec["loc_noise"] * torch.randn(flow.base_distribution.loc.shape,
                                                                            generator=torch.Generator().manual_seed(spec["seed"])))
    return flow


def base_state(flow):
    return {k: v.detach().clone() for k, v in flow.state_dict().items() if k.startswith("base_distribution.")}


def relu_margin(flow, x):
    """smallest |input| of any (Leaky)ReLU of the flow on this batch, relative to that input tensor's largest entry: the
    derivative of a ReLU whose input lies within fp32 rounding of zero (~1e-6) is either one-sided value, depending on the
    order of evaluation -- a gradient fixture must stay clear of that"""
    worst = [float("inf")]

    def hook(_m, inp, _out):
        t = inp[0].detach()
        worst[0] = min(worst[0], (t.abs().min() / t.abs().max().clamp_min(1e-30)).item())

    hs = [m.register_forward_hook(hook) for m in flow.modules() if isinstance(m, (torch.nn.ReLU, torch.nn.LeakyReLU))]
    try:
        with torch.no_grad():
            flow.log_prob(x)
    finally:
        for h in hs:
            h.remove()
    return worst[0]


def run_case(name, in_dims, K, base, seed, n=12, p=1.0, loc_noise=0.0, grad_layers=None, kink_margin=None):
    """kink_margin: draw the inputs from the first generator seed 1000 + seed + 100 j (j = 0, 1, ...) for which every ReLU
    input of the fp32 reference stays at least that far (relative) from zero (round 5: the seed-51 inputs of the CIFAR
    k10 case held a pre-activation of 1.4e-7 -- the device's gradient then depends on its order of additions)"""
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        return
    spec = dict(in_dims=list(in_dims), coupling_blocks=K, base=base, seed=seed, p=p, loc_noise=loc_noise, cond_args=LIVE_COND,
                prior_scale=1.0)
    flow = build(spec)
    bsd = base_state(flow)
    x_seed = 1000 + seed
    while True:
        g = torch.Generator().manual_seed(x_seed)
        x = torch.rand(n, *in_dims, generator=g)
        margin = relu_margin(flow, x)
        if kink_margin is None or margin >= kink_margin:
            break
        print(f"{name}: input seed {x_seed}: smallest relative |ReLU input| {margin:.2e} < {kink_margin:.0e}; next seed")
        x_seed += 100
    print(f"{name}: input seed {x_seed}, smallest relative |ReLU input| {margin:.2e}")
    spec["x_seed"] = x_seed
    # (latents for _forward: the reference's RadialDistribution.sample raises a shape error for an image-shaped loc --
    # distributions.py:482-494 repeats the [n, 1] radii along the wrong axes -- so they are drawn here directly)
    zin = 0.5 * torch.randn(n, *in_dims, generator=g)
    out = {}
    with torch.no_grad():
        out["log_prob32"], out["backward32"], out["forward32"] = flow.log_prob(x), flow.backward(x), flow._forward(zin)
        out["base_log_prob32"] = flow.base_distribution.log_prob(out["backward32"])
    torch.set_default_dtype(torch.float64)
    try:
        f64 = flow.double()
        for l in f64.layers:
            if isinstance(l, transforms.MaskedCoupling):
                l.mask = l.mask.double()
        with torch.no_grad():
            out["log_prob64"], out["backward64"], out["forward64"] = f64.log_prob(x.double()), f64.backward(x.double()), f64._forward(zin.double())
            out["base_log_prob64"] = f64.base_distribution.log_prob(out["backward64"])
            ladj = 0.0
            for l in f64.layers:
                ladj = ladj + l.log_abs_det_jacobian(None, None)
            out["total_ladj64"] = torch.as_tensor(ladj, dtype=torch.float64)
        for q in f64.parameters():
            q.grad = None
        lp = f64.log_prob(x.double())
        prior = f64.log_prior()
        loss = -lp.mean() - prior
        loss.backward()
        out["loss64"] = loss.detach()
        out["log_prior64"] = torch.as_tensor(float(prior), dtype=torch.float64)
        grads = {k: q.grad.detach().clone() for k, q in f64.named_parameters() if q.grad is not None}
        if grad_layers is not None:
            # a deep flow: the gradients of the base and of the listed trainable layers only (head, middle, tail of the
            # chain) keep the fixture small; every gradient still depends on the whole backward chain behind it
            keep = tuple(f"trainable_layers.{i}." for i in grad_layers)
            grads = {k: v for k, v in grads.items() if k.startswith("base_distribution.") or k.startswith(keep)}
    finally:
        torch.set_default_dtype(torch.float32)
    arrays = {"x": x.numpy(), "zin": zin.numpy()}
    arrays.update({k: v.detach().numpy() for k, v in out.items()})
    arrays.update({"sd/" + k: v.float().numpy() for k, v in bsd.items()})
    arrays.update({"g/" + k: v.numpy() for k, v in grads.items()})
    arrays["spec"] = np.array(json.dumps(spec))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    rel = (out["log_prob32"].double() - out["log_prob64"]).abs() / out["log_prob64"].abs()
    r = out["backward64"].flatten(1).abs().sum(-1)
    print(f"{name:56s} logp[0]={out['log_prob64'][0].item():+.6e} ref32-vs-64 {rel.max().item():.2e} r in [{r.min().item():.4g}, "
          f"{r.max().item():.4g}] prior {float(prior):+.3g} {len(grads)} grads {os.path.getsize(path) / 1024:.0f} KB")


def fit_case(name, in_dims, K, base, seed):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        return
    spec = dict(in_dims=list(in_dims), coupling_blocks=K, base=base, seed=seed, p=1.0, loc_noise=0.0, cond_args=LIVE_COND,
                prior_scale=1.0)
    flow = build(spec)
    bsd = base_state(flow)
    data = torch.rand(N_ROWS, *in_dims, generator=torch.Generator().manual_seed(77))
    ds = torch.utils.data.TensorDataset(data, torch.zeros(N_ROWS))
    np.random.seed(NP_SEED)
    # (default optimiser = SophiaG, flows.py:116; the live hyper-parameters mnist.yaml:36-42)
    losses = flow.fit(ds, optim_params=dict(lr=LR, weight_decay=0.0), batch_size=BATCH, shuffle=True, device=torch.device("cpu"),
                      epochs=EPOCHS)
    arrays = {"losses": np.array(losses, dtype=np.float64), "data": data.numpy(), "spec": np.array(json.dumps(spec))}
    arrays.update({"sd0/" + k: v.numpy() for k, v in bsd.items()})
    for k, v in flow.state_dict().items():
        arrays["sd/" + k] = v.detach().numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name:56s} epoch losses {losses}  {os.path.getsize(path) / 1024:.0f} KB")


def main():
    run_case("imageradial_mnistlive_c16_7x7_k2_l3_lognormal", (16, 7, 7), 2, "lognormal", 41)
    run_case("imageradial_mnistlive_c16_7x7_k2_l3_lognormal_loc", (16, 7, 7), 2, "lognormal", 42, loc_noise=0.05)
    run_case("imageradial_fashionlive_c16_7x7_k2_l3_gammamm", (16, 7, 7), 2, "gammamm", 43)
    run_case("imageradial_cifarlive_c48_8x8_k2_l3_lognormal", (48, 8, 8), 2, "lognormal", 44, n=6)
    # the live depth: 15 coupling blocks x 3 gated layers (mnist.yaml:56-72)
    run_case("imageradial_mnistlive_c16_7x7_k15_l3_lognormal", (16, 7, 7), 15, "lognormal", 45, n=8, grad_layers=(0, 1, 14, 15, 29, 30, 31))
    # the other norms of the radial base (p = 2, inf: distributions.py:513-549) on an image-shaped event, and the Fashion / CIFAR
    # configurations at their live depth of 10 blocks (gradients of the base and of the head / middle / tail layers)
    run_case("imageradial_c16_7x7_k2_l3_lognormal_p2", (16, 7, 7), 2, "lognormal", 48, n=6, p=2.0, loc_noise=0.05, grad_layers=(0, 1, 4, 5))
    run_case("imageradial_c16_7x7_k2_l3_gammamm_pinf", (16, 7, 7), 2, "gammamm", 49, n=6, p=float("inf"), loc_noise=0.05,
             grad_layers=(0, 1, 4, 5))
    run_case("imageradial_fashionlive_c16_7x7_k10_l3_gammamm", (16, 7, 7), 10, "gammamm", 50, n=6, grad_layers=(0, 1, 10, 19, 20, 21))
    run_case("imageradial_cifarlive_c48_8x8_k10_l3_lognormal", (48, 8, 8), 10, "lognormal", 51, n=4, grad_layers=(0, 1, 10, 19, 20, 21),
             kink_margin=2e-6)
    fit_case("imageradialfit_mnistlive_c16_7x7_k2_l3_lognormal", (16, 7, 7), 2, "lognormal", 46)
    fit_case("imageradialfit_fashionlive_c16_7x7_k2_l3_gammamm", (16, 7, 7), 2, "gammamm", 47)


if __name__ == "__main__":
    main()
