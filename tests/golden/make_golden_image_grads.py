"""Golden GRADIENTS and a golden ``Flow.fit`` run of IMAGE-SHAPED flows from the REAL reference (this container only):
what the device training path of image flows (usflows_amd/image_training.py) is checked against.

    python tests/golden/make_golden_image_grads.py     # writes tests/golden/imagegrads_<case>.npz, imagefit_<case>.npz

Gradients: for image cases of make_golden_image.py the reference ``USFlow`` is rebuilt exactly as there (same constructor
call, same parameters: the stored state dict, or tests/image_synth.py for the *_synth cases), moved to fp64, and the
gradient of the training loss of ``Flow.fit`` (flows.py:196-199: ``-log_prob(x).mean()``) w.r.t. every parameter is stored.
Fit: two epochs of plain SGD over 96 rows (batch 32, shuffle=True under a fixed numpy seed) of the MNIST experiment
model (tests/explib/mnist.yaml:44-77), fp32 on the CPU as the reference runs it; stored: the per-epoch losses and every
parameter after the 6 steps.  Data only."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_golden_image as mgi  # noqa: E402  (imports the reference through ref_shim; main() is not run)

flows, transforms, networks, distributions = mgi.flows, mgi.transforms, mgi.networks, mgi.distributions

GRAD_CASES = ["image_mnistcfg_c16_7x7_k2_gated_ln_hh1_conj", "image_mnistcfg_c16_7x7_k2_gated_ln_hh1_conj_synth",
              "image_cifarcfg_c48_8x8_k2_gated_ln_hh1_conj", "image_c16_7x7_k2_plain_channelmask"]
FIT_CASE = "image_mnistcfg_c16_7x7_k2_gated_ln_hh1_conj"
LR, NP_SEED, N_ROWS, BATCH, EPOCHS = 1e-3, 5, 96, 32, 2


def build(name):
    """the reference flow of a stored image case with its parameters, and the case's arrays"""
    a = np.load(os.path.join(HERE, name + ".npz"))
    spec = json.loads(str(a["spec"]))
    in_dims = spec["in_dims"]
    base = torch.distributions.Laplace(torch.zeros(in_dims), torch.ones(in_dims))
    flow = flows.USFlow(base, list(in_dims), spec["coupling_blocks"], networks.ConvNet2D, dict(spec["cond_args"]),
                        householder=spec["householder"], affine_conjugation=spec["affine_conjugation"],
                        masktype=spec["masktype"])
    if "synth_seed" in spec:
        from image_synth import synth_image_params_
        synth_image_params_(flow, spec["synth_seed"])
    else:
        sd = {k[3:]: torch.from_numpy(a[k]) for k in a.files if k.startswith("sd/")}
        res = flow.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
    return flow, spec, a


def grads(name):
    flow, spec, a = build(name)
    in_dims = spec["in_dims"]
    torch.set_default_dtype(torch.float64)
    try:
        f64 = flow.double()
        for l in f64.layers:
            if isinstance(l, transforms.MaskedCoupling):
                l.mask = l.mask.double()
        f64.base_distribution = distributions.Independent(
            torch.distributions.Laplace(torch.zeros(in_dims).double(), torch.ones(in_dims).double()), len(in_dims))
        x = torch.from_numpy(a["x"]).double()
        for p in f64.parameters():
            p.grad = None
        lp = f64.log_prob(x)
        assert float((lp.detach() - torch.from_numpy(a["log_prob64"])).abs().max()) < 1e-9, "not the flow of the stored case"
        (-lp.mean()).backward()
        arrays = {"loss": np.array(float(-lp.mean()))}
        for k, p in f64.named_parameters():
            if p.grad is not None:
                arrays["g/" + k] = p.grad.detach().numpy()
    finally:
        torch.set_default_dtype(torch.float32)
    path = os.path.join(HERE, "imagegrads_" + name[len("image_"):] + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name:52s} loss {arrays['loss']:+.6e}  {len(arrays) - 1} gradients  {os.path.getsize(path) / 1024:.0f} KB")


def fit(name):
    flow, spec, a = build(name)
    in_dims = spec["in_dims"]
    data = torch.rand(N_ROWS, *in_dims, generator=torch.Generator().manual_seed(77))
    ds = torch.utils.data.TensorDataset(data, torch.zeros(N_ROWS))
    np.random.seed(NP_SEED)
    losses = flow.fit(ds, optim=torch.optim.SGD, optim_params=dict(lr=LR), batch_size=BATCH, shuffle=True,
                      device=torch.device("cpu"), epochs=EPOCHS)
    arrays = {"losses": np.array(losses, dtype=np.float64), "data": data.numpy()}
    for k, v in flow.state_dict().items():
        arrays["sd/" + k] = v.detach().numpy()
    path = os.path.join(HERE, "imagefit_" + name[len("image_"):] + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name:52s} epoch losses {losses}  {os.path.getsize(path) / 1024:.0f} KB")


def main():
    for name in GRAD_CASES:
        grads(name)
    fit(FIT_CASE)


if __name__ == "__main__":
    main()
