"""Golden vectors of the reference's LIVE FLAT configuration from the REAL reference (this container only).

    python tests/golden/make_golden_gaussian_mixture.py   # writes tests/golden/init_d<D>_k10_gmlive.npz, gmfit_d<D>_k10_gmlive.npz

What ``HyperoptExperiment._trial`` builds and trains for experiments/synthetic/gaussian_mixture.yaml:50-93 (and its 3-D ... 100-D
overwrites, :94-420): ``USFlow(in_dims=[D], coupling_blocks=10, conditioner_cls=pyro.nn.DenseNN, conditioner_args=dict(input_dim=D,
hidden_dims=[32, 32], param_dims=[D]), nonlinearity=ReLU(), lu_transform=1, householder=0, affine_conjugation=True,
soft_training=False, training_noise_prior=Uniform(1e-20, 0.01), prior_scale=1.0, base_distribution=RadialDistribution(p=1,
loc=zeros[D], norm_distribution=GammaMM(concentration=rand[20] * sqrt(D), rate=ones[20], mixture_weights=ones[20] / 20)))``,
batch 32, SophiaG lr 1e-3 weight_decay 0 -- constructor call for constructor call, for D = 2, 10, 100.
(pyro.nn.DenseNN itself is third-party source that is not under /root/reference: tests/golden/ref_shim.py restates its published
algorithm -- the one boundary of this fixture that is pinned by the restatement, not by the reference)

Parameters: what the reference's own constructors draw under ``torch.manual_seed(seed)`` -- the initialisation a trial starts
from -- for D = 2 and 10; at D = 100 the default initialisation is badly conditioned through 21 affine layers, so
that case applies the documented conditioning transform of oracle/synth.py to the LU factors (everything else as drawn).
Per case: the state dict, inputs, ``log_prob`` / ``backward`` / ``_forward`` in fp32 and fp64, the log-det total, ``log_prior()``
and the fp64 gradient of ``Flow.fit``'s loss ``-log_prob(x).mean() - log_prior()`` (flows.py:196-198) w.r.t. EVERY parameter --
the GammaMM's concentration / rate / mixture weights and the radial ``loc`` included.  Fit cases: ``Flow.fit`` with its default
optimiser SophiaG at the live hyper-parameters, 2 epochs x 96 rows, batch 32, fp32 on the CPU: per-epoch losses and every
parameter after the 6 steps.  Data only."""
import json
import math
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
from pyro.nn import DenseNN  # noqa: E402  (the shim's restatement)

LR, NP_SEED, N_ROWS, BATCH, EPOCHS = 1e-3, 5, 96, 32, 2


def build(D, seed, condition):
    torch.manual_seed(seed)
    nd = distributions.GammaMM(concentration=torch.rand([20]) * torch.sqrt(torch.tensor([float(D)])), rate=torch.ones([20]),
                               mixture_weights=torch.ones([20]) / 20, device="cpu")
    base = distributions.RadialDistribution(device="cpu", p=float("1"), loc=torch.zeros([D]), norm_distribution=nd)
    flow = flows.USFlow(base_distribution=base, in_dims=[D], coupling_blocks=10, conditioner_cls=DenseNN,
                        conditioner_args=dict(input_dim=D, hidden_dims=[32, 32], param_dims=[D]), soft_training=False,
                        training_noise_prior=torch.distributions.Uniform(1e-20, 0.01), prior_scale=1.0, lu_transform=1,
                        householder=0, affine_conjugation=True, nonlinearity=torch.nn.ReLU())
    if condition:
        # SURVEY 7-H2 / oracle/synth.py: L <- I + alpha tril(L, -1); U <- alpha triu(U, 1) + diag(sign U[0.75, 1.25])
        g = torch.Generator().manual_seed(seed)
        alpha = 0.1
        with torch.no_grad():
            for name, p in flow.named_parameters():
                if name.endswith("L_raw"):
                    p.copy_(torch.eye(D) + alpha * p.tril(-1))
                elif name.endswith("U_raw"):
                    sign = torch.where(p.diagonal() < 0, -1.0, 1.0)
                    p.copy_(alpha * p.triu(1) + torch.diag(sign * (0.75 + 0.5 * torch.rand(D, generator=g))))
    return flow


def spec_json(D, seed, condition):
    return json.dumps(dict(dim=D, coupling_blocks=10, hidden_dims=[32, 32], lu_transform=1, householder=0, affine_conjugation=True,
                           negative_slope=0.0, conditioner="DenseNN", base="radial", radial_p=1.0, radial_norm="gammamm",
                           radial_norm_loc=0.0, radial_norm_scale=1.0, soft_training=False,
                           extra={"gammamm_k": 20, "prior_scale": 1.0, "conditioned": bool(condition)}))


def data_rows(D, n, seed):
    # rows like the experiment's data set: a two-component Gaussian mixture at -1 / +1 (gaussian_mixture.yaml:16-29)
    g = torch.Generator().manual_seed(1000 + seed)
    comp = (torch.rand(n, generator=g) < 0.5).float()[:, None] * 2 - 1
    return comp * torch.ones(D) + 0.5 * torch.randn(n, D, generator=g)


def run_case(D, seed, condition, n=32):
    name = f"init_d{D}_k10_gmlive"
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        return
    flow = build(D, seed, condition)
    sd = {k: v.detach().clone() for k, v in flow.state_dict().items()}
    x = data_rows(D, n, seed)
    zin = 0.7 * torch.randn(n, D, generator=torch.Generator().manual_seed(2000 + seed))
    out = {}
    with torch.no_grad():
        out["log_prob32"], out["backward32"], out["forward32"] = flow.log_prob(x), flow.backward(x), flow._forward(zin)
    torch.set_default_dtype(torch.float64)
    try:
        f64 = flow.double()
        for l in f64.layers:
            if isinstance(l, transforms.MaskedCoupling):
                l.mask = l.mask.double()
        with torch.no_grad():
            out["log_prob64"], out["backward64"], out["forward64"] = f64.log_prob(x.double()), f64.backward(x.double()), f64._forward(zin.double())
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
    finally:
        torch.set_default_dtype(torch.float32)
    arrays = {"x": x.numpy(), "zin": zin.numpy()}
    arrays.update({k: v.detach().numpy() for k, v in out.items()})
    arrays.update({"sd/" + k: v.float().numpy() for k, v in sd.items()})
    arrays.update({"g/" + k: v.numpy() for k, v in grads.items()})
    arrays["spec"] = np.array(spec_json(D, seed, condition))
    arrays["family"] = np.array("init")
    arrays["seed"] = np.array(seed)
    arrays["alpha"] = np.array(0.1)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    rel = (out["log_prob32"].double() - out["log_prob64"]).abs() / out["log_prob64"].abs()
    print(f"{name:28s} logp[0]={out['log_prob64'][0].item():+.6e} ref32-vs-64 {rel.max().item():.2e} max|z| "
          f"{out['backward64'].abs().max().item():.3g} prior {float(prior):+.3g} {len(grads)} grads {os.path.getsize(path) / 1024:.0f} KB")


def fit_case(D, seed, condition):
    name = f"gmfit_d{D}_k10_gmlive"
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        return
    flow = build(D, seed, condition)
    sd0 = {k: v.detach().clone() for k, v in flow.state_dict().items()}
    data = data_rows(D, N_ROWS, seed + 7)
    ds = torch.utils.data.TensorDataset(data, torch.zeros(N_ROWS))
    np.random.seed(NP_SEED)
    # (default optimiser = SophiaG, flows.py:116; the live hyper-parameters gaussian_mixture.yaml:40-49)
    losses = flow.fit(ds, optim_params=dict(lr=LR, weight_decay=0.0), batch_size=BATCH, shuffle=True, device=torch.device("cpu"),
                      epochs=EPOCHS)
    arrays = {"losses": np.array(losses, dtype=np.float64), "data": data.numpy(), "spec": np.array(spec_json(D, seed, condition))}
    arrays.update({"sd0/" + k: v.numpy() for k, v in sd0.items()})
    for k, v in flow.state_dict().items():
        arrays["sd/" + k] = v.detach().numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name:28s} epoch losses {losses}  {os.path.getsize(path) / 1024:.0f} KB")


def main():
    for D, seed, condition in ((2, 61, False), (10, 62, False), (100, 63, True)):
        run_case(D, seed, condition)
    for D, seed, condition in ((2, 64, False), (10, 65, False)):
        fit_case(D, seed, condition)


if __name__ == "__main__":
    main()
