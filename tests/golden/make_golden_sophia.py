"""Golden vectors of the REAL reference's SophiaG (sophia.py) and of ``Flow.fit`` with its DEFAULT optimiser
(flows.py:116: SophiaG; update_hessian is never called by fit, so the Hessian estimate stays zero and a step is
weight decay + lr * sign(momentum)) -- this container only.

    python tests/golden/make_golden_sophia.py     # writes tests/golden/sophia_steps.npz, tests/golden/fitsophia_<case>.npz

sophia_steps.npz: three fp32 tensors, 6 steps with fresh random gradients, update_hessian() before steps 0 and 3,
step(bs=7); a second run with maximize=True.  Stored: initial tensors, gradients, and parameters / exp_avg / hessian after
every step.  Data only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_golden as mg  # noqa: E402  (installs the plumbing-only pyro shim and imports the reference)
from golden_util import load_case  # noqa: E402
from src.usflows.sophia import SophiaG  # noqa: E402

SHAPES = [(5, 7), (13,), (1,), (64, 33)]
STEPS, BS, HESS_AT = 6, 7, (0, 3)
HYPER = dict(lr=3e-3, betas=(0.9, 0.95), rho=0.05, weight_decay=0.2)
FIT_CASE, NP_SEED, N_ROWS, BATCH, EPOCHS = "synth_d7_k3_hh0_laplace", 5, 96, 32, 2


def run(maximize):
    g = torch.Generator().manual_seed(11)
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in SHAPES]
    out = {f"p0/{i}": p.detach().numpy().copy() for i, p in enumerate(ps)}
    opt = SophiaG(ps, maximize=maximize, **HYPER)
    for t in range(STEPS):
        for i, p in enumerate(ps):
            p.grad = torch.randn(*SHAPES[i], generator=g) * (10.0 ** (t - 3))      # magnitudes on both sides of the clamp
            out[f"g/{t}/{i}"] = p.grad.numpy().copy()
        if t in HESS_AT:
            opt.update_hessian()
        opt.step(bs=BS)
        for i, p in enumerate(ps):
            st = opt.state[p]
            out[f"p/{t}/{i}"] = p.detach().numpy().copy()
            out[f"m/{t}/{i}"] = st["exp_avg"].numpy().copy()
            out[f"h/{t}/{i}"] = st["hessian"].numpy().copy()
    return out


def main():
    arrays = {}
    for mx in (False, True):
        for k, v in run(mx).items():
            arrays[f"max{int(mx)}/{k}"] = v
    path = os.path.join(HERE, "sophia_steps.npz")
    np.savez_compressed(path, **arrays)
    print(f"sophia_steps.npz  {os.path.getsize(path) / 1024:.0f} KB")

    spec, sd, a = load_case(FIT_CASE)
    seed = int(np.load(os.path.join(HERE, FIT_CASE + ".npz"))["seed"])
    flow = mg.build_reference(spec, seed)
    res = flow.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    data = torch.rand(N_ROWS, spec.dim, generator=torch.Generator().manual_seed(77))
    ds = torch.utils.data.TensorDataset(data, torch.zeros(N_ROWS))
    np.random.seed(NP_SEED)
    losses = flow.fit(ds, batch_size=BATCH, shuffle=True, device=torch.device("cpu"), epochs=EPOCHS)    # default optimiser
    arrays = {"losses": np.array(losses, dtype=np.float64), "data": data.numpy()}
    for k, v in flow.state_dict().items():
        arrays["sd/" + k] = v.detach().numpy()
    path = os.path.join(HERE, "fitsophia_" + FIT_CASE + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"fitsophia_{FIT_CASE}: epoch losses {losses}  {os.path.getsize(path) / 1024:.0f} KB")


if __name__ == "__main__":
    main()
