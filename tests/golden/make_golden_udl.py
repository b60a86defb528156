"""Golden radial UDL profiles from the REAL reference (this container only):
``Flow.calibrated_latent_radial_udl_profile`` (flows.py:294-378) with ``RadialDistribution.radial_udl_profile`` /
``radial_ldl_profile`` (distributions.py:390-456) on the radial golden cases.

    python tests/golden/make_golden_udl.py        # writes tests/golden/udl_<case>.npz   (data only)
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

CASES = ["synth_d16_k3_hh1_radial2", "synth_d16_k4_hh0_conj_radial1", "synth_d16_k3_hh0_radialinf"]
Q, R_MAX, N_SAMPLES = 0.8, 60.0, 6000


def main():
    for name in CASES:
        spec, sd, a = load_case(name)
        seed = int(np.load(os.path.join(HERE, name + ".npz"))["seed"])
        flow = mg.build_reference(spec, seed)
        res = flow.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys
        out = {}
        for cut in (True, False):
            prof = flow.calibrated_latent_radial_udl_profile(Q, a["x"], r_max=R_MAX, n_samples=N_SAMPLES,
                                                             cut_to_data_tail=cut)
            out["cut" if cut else "full"] = prof.detach().numpy().astype(np.float64)
        with torch.no_grad():
            r = (flow.backward(a["x"]) - flow.base_distribution.loc).norm(p=spec.radial_p, dim=1)
        path = os.path.join(HERE, "udl_" + name + ".npz")
        np.savez_compressed(path, q=np.array(Q), r_max=np.array(R_MAX), n_samples=np.array(N_SAMPLES),
                            latent_radius=r.numpy(), **out)
        print(f"{name:34s} cut {out['cut'].tolist()}  full {out['full'].tolist()}")


if __name__ == "__main__":
    main()
