"""The image-shape oracle (oracle/usflows_image_oracle.py) pinned by the golden vectors of the REAL reference for image
flows (tests/golden/image_*.npz; tests/golden/make_golden_image.py): fp64 and fp32 runs of seven configurations, the
reference's MNIST and CIFAR ones (the latter with all 10 blocks) among them."""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR, image_case_names, load_image_case
from oracle import usflows_image_oracle as iorc
from oracle.usflows_oracle import to_dtype


def _spec_and_sd(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    d = json.loads(str(z["spec"]))
    spec = iorc.ImageSpec(in_dims=d["in_dims"], coupling_blocks=d["coupling_blocks"], cond_args=dict(d["cond_args"]),
                          householder=d["householder"], affine_conjugation=d["affine_conjugation"], masktype=d["masktype"])
    if "synth_seed" in d:
        flow, _ = load_image_case(name)                 # parameters regenerated from the seed; the state dict is data
        sd = {k: v.detach().clone() for k, v in flow.state_dict().items()}
    else:
        sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("sd/") and k != "spec"}
    return spec, sd, arrays


def _rel(a, b):
    return ((a.double() - b.double()).abs() / b.double().abs().clamp_min(1e-30)).max().item()


@pytest.mark.parametrize("name", image_case_names())
def test_image_oracle_matches_reference_fp64_and_fp32(name):
    spec, sd, a = _spec_and_sd(name)
    sd64 = to_dtype(sd, torch.float64)
    with torch.no_grad():
        lp64 = iorc.flow_log_prob(sd64, spec, a["x"].double())
        z64 = iorc.flow_backward(sd64, spec, a["x"].double())
        x64 = iorc.flow_forward(sd64, spec, a["zin"].double())
        lp32 = iorc.flow_log_prob(sd, spec, a["x"])
        z32 = iorc.flow_backward(sd, spec, a["x"])
    assert _rel(lp64, a["log_prob64"]) < 1e-11
    assert (z64 - a["backward64"]).abs().max().item() < 1e-10 * max(1.0, a["backward64"].abs().max().item())
    assert (x64 - a["forward64"]).abs().max().item() < 1e-10 * max(1.0, a["forward64"].abs().max().item())
    assert abs(float(iorc.total_ladj(sd64, spec)) - float(a["total_ladj64"])) < 1e-9 * max(1.0, abs(float(a["total_ladj64"])))
    # the same torch ops in the same order as the reference's fp32 run
    assert _rel(lp32, a["log_prob32"]) < 2e-6
    assert (z32 - a["backward32"]).abs().max().item() < 2e-5 * max(1.0, a["backward32"].abs().max().item())


def test_image_masks_as_the_reference_builds_them():
    spec = iorc.ImageSpec(in_dims=[3, 4, 5], coupling_blocks=1)
    m = iorc.image_mask(spec, 0)
    assert m.shape == (1, 3, 4, 5)
    for c in range(3):
        for h in range(4):
            for w in range(5):
                assert m[0, c, h, w].item() == (c + h + w) % 2                 # flows.py:506-510
    assert torch.equal(iorc.image_mask(spec, 1), 1 - m)
    spec.masktype = "channel"
    mc = iorc.image_mask(spec, 0)
    assert all(torch.all(mc[0, c] == c % 2) for c in range(3))                 # flows.py:530-532
