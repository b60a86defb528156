"""Golden run of the REAL reference's ``Flow.fit`` (this container only): pins the training LOOP the device path
plugs into (flows.py:113-210: shuffling, batching, loss = -log_prob(batch).mean() - log_prior(), optimiser step).

    python tests/golden/make_golden_fit.py        # writes tests/golden/fit_<case>.npz

Two epochs of plain SGD over 96 rows (batch 32, shuffle=True under a fixed numpy seed) starting from the stored
state dict of a small golden case; stored: the per-epoch losses and every parameter after the 6 steps.  Data only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_golden as mg  # noqa: E402
from golden_util import load_case  # noqa: E402

# (case, prior_scale): with a prior scale the loss carries -log_prior() (LUTransform.log_prior, transforms.py:1371-1379)
CASES = [("synth_d7_k3_hh0_laplace", None), ("synth_d16_k3_hh1_radial2", None), ("synth_d7_k3_hh1_conj_normal", 0.5),
         ("synth_d16_k3_convnet_gated_ln", None)]      # (round 5: the vector ConvNet conditioner with GatedMLP / LayerNormVector blocks)
LR, NP_SEED, N_ROWS, BATCH, EPOCHS = 1e-3, 5, 96, 32, 2


def main():
    for name, prior_scale in CASES:
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        spec, sd, a = load_case(name)
        seed = int(np.load(os.path.join(HERE, name + ".npz"))["seed"])
        flow = mg.build_reference(spec, seed)
        if prior_scale is not None:
            # same constructor call as make_golden.build_reference, plus the prior scale (flows.py:397)
            act = torch.nn.LeakyReLU(spec.negative_slope)
            flow = mg.flows.USFlow(mg.make_base(spec), [spec.dim], spec.coupling_blocks, mg.networks.ConditionalDenseNN,
                                   dict(input_dim=spec.dim, context_dim=1, hidden_dims=list(spec.hidden_dims),
                                        out_dim=spec.dim, nonlinearity=act),
                                   affine_conjugation=spec.affine_conjugation, lu_transform=spec.lu_transform,
                                   householder=spec.householder, prior_scale=prior_scale)
        res = flow.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys
        data = torch.rand(N_ROWS, spec.dim, generator=torch.Generator().manual_seed(77))
        ds = torch.utils.data.TensorDataset(data, torch.zeros(N_ROWS))
        np.random.seed(NP_SEED)
        losses = flow.fit(ds, optim=torch.optim.SGD, optim_params=dict(lr=LR), batch_size=BATCH, shuffle=True,
                          device=torch.device("cpu"), epochs=EPOCHS)
        arrays = {"losses": np.array(losses, dtype=np.float64), "data": data.numpy(),
                  "prior_scale": np.array(-1.0 if prior_scale is None else prior_scale)}
        for k, v in flow.state_dict().items():
            arrays["sd/" + k] = v.detach().numpy()
        path = os.path.join(HERE, "fit_" + name + ".npz")
        np.savez_compressed(path, **arrays)
        print(f"{name:36s} epoch losses {losses}  {os.path.getsize(path) / 1024:.0f} KB")


if __name__ == "__main__":
    main()
