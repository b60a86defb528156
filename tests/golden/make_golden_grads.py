"""Golden GRADIENTS from the REAL reference (this container only): pins the oracle's autograd, which the device
training path (usflows_amd/training.py) is checked against.

    python tests/golden/make_golden_grads.py      # writes tests/golden/grads_<case>.npz

For a handful of the small cases of make_golden.py: the stored state dict is loaded into the reference ``USFlow``
(fp64, through the plumbing-only pyro shim), and the gradient of the training loss of ``Flow.fit``
(flows.py:196-199: ``-log_prob(x, context).mean()``) w.r.t. every parameter is stored.  Data only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_golden as mg  # noqa: E402  (imports the reference through ref_shim; main() is not run)
from golden_util import load_case  # noqa: E402

CASES = ["synth_d7_k3_hh0_laplace", "synth_d7_k3_hh1_conj_normal", "synth_d16_k4_hh2_conj_laplace",
         "synth_d16_k3_hh1_radial2", "synth_d16_k4_hh0_conj_radial1", "synth_d16_k3_hh0_radialinf",
         "synth_d7_k3_soft_ctx", "synth_d16_k3_densenn_relu", "synth_d33_k3_lu2_hh1",
         # round 5: the vector ConvNet conditioner (GatedMLP / LayerNormVector blocks, networks.py:206-245, 287-308) under autograd
         "synth_d16_k3_convnet_gated_ln", "synth_d33_k2_convnet_gated_conj", "synth_d64_k3_convnet_ln", "init_d4_k2_convnet_default"]


def main():
    for name in CASES:
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        spec, sd, a = load_case(name)
        seed = int(np.load(os.path.join(HERE, name + ".npz"))["seed"])
        flow = mg.build_reference(spec, seed)
        res = flow.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
        torch.set_default_dtype(torch.float64)
        try:
            flow64 = mg.to_double(flow, spec)
            if spec.base == "radial":                  # the trainable loc of the stored state dict, in fp64
                flow64.base_distribution.loc.data = sd["base_distribution.loc"].double()
            x = a["x"].double()
            ctx = a["context"].double() if "context" in a else None
            for p in flow64.parameters():
                p.grad = None
            lp = flow64.log_prob(x, context=ctx) if ctx is not None else flow64.log_prob(x)
            (-lp.mean()).backward()
            arrays = {"loss": np.array(float(-lp.mean()))}
            for k, p in flow64.named_parameters():
                if p.grad is not None:
                    arrays["g/" + k] = p.grad.detach().numpy()
        finally:
            torch.set_default_dtype(torch.float32)
        path = os.path.join(HERE, "grads_" + name + ".npz")
        np.savez_compressed(path, **arrays)
        print(f"{name:40s} loss {arrays['loss']:+.6e}  {len(arrays) - 1} gradients  {os.path.getsize(path) / 1024:.0f} KB")


if __name__ == "__main__":
    main()
