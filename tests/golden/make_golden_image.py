"""Golden vectors for IMAGE-SHAPED flows (first slice) from the REAL reference (this container only).

    python tests/golden/make_golden_image.py        # rewrites tests/golden/image_*.npz

Reference ``USFlow(in_dims=[C, H, W], conditioner_cls=ConvNet2D, ...)`` (flows.py:389-491: 1x1-conv
``BlockAffineTransform`` transforms.py:904-962, image checkerboard / channel masks flows.py:494-536, CNN conditioner
networks.py:405-510) built under ``torch.manual_seed``, its LU / scale parameters conditioned as in SURVEY 7-H2 (the
default init explodes), then run in fp32 and fp64.  Stored: the state dict (small), inputs, outputs.  Data only."""
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))          # tests/: image_synth
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_shim  # noqa: E402

flows, transforms, networks, distributions = ref_shim.install()


def condition_(flow, seed, alpha=0.3):
    """L <- I + alpha tril(L,-1); U <- alpha triu(U,1) + diag(sign U[0.75,1.25]); scale <- sign U[0.5,1.5]"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in flow.modules():
            if isinstance(m, transforms.LUTransform):
                D = m.L_raw.shape[0]
                m.L_raw.copy_(torch.eye(D) + alpha * m.L_raw.tril(-1))
                sign = torch.where(torch.rand(D, generator=g) < 0.5, -1.0, 1.0)
                m.U_raw.copy_(alpha * m.U_raw.triu(1) + torch.diag(sign * (0.75 + 0.5 * torch.rand(D, generator=g))))
            if isinstance(m, transforms.ScaleTransform):
                sign = torch.where(torch.rand(m.scale.shape, generator=g) < 0.5, -1.0, 1.0)
                m.scale.copy_(sign * (0.5 + torch.rand(m.scale.shape, generator=g)))


def run_case(name, in_dims, K, cond_args, seed, hh=1, conj=True, masktype="checkerboard", n=12, synth=False):
    """synth: the parameters come from tests/image_synth.py (a pure function of the seed and the module structure, which
    the mirror shares key for key): no state dict is stored, the test regenerates it"""
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        return
    torch.manual_seed(seed)
    base = torch.distributions.Laplace(torch.zeros(in_dims), torch.ones(in_dims))
    flow = flows.USFlow(base, list(in_dims), K, networks.ConvNet2D, dict(cond_args), householder=hh,
                        affine_conjugation=conj, masktype=masktype)
    if synth:
        from image_synth import synth_image_params_
        synth_image_params_(flow, seed)
        sd = {}
    else:
        condition_(flow, seed)
        sd = {k: v.detach().clone() for k, v in flow.state_dict().items()}
    g = torch.Generator().manual_seed(1000 + seed)
    x = torch.rand(n, *in_dims, generator=g)
    zin = torch.distributions.Laplace(0.0, 1.0).icdf(torch.rand(n, *in_dims, generator=g) * 0.998 + 0.001)
    out = {}
    with torch.no_grad():
        out["log_prob32"], out["backward32"], out["forward32"] = flow.log_prob(x), flow.backward(x), flow._forward(zin)
        torch.set_default_dtype(torch.float64)
        f64 = flow.double()
        for l in f64.layers:
            if isinstance(l, transforms.MaskedCoupling):
                l.mask = l.mask.double()
        f64.base_distribution = distributions.Independent(
            torch.distributions.Laplace(torch.zeros(in_dims).double(), torch.ones(in_dims).double()), len(in_dims))
        out["log_prob64"], out["backward64"], out["forward64"] = f64.log_prob(x.double()), f64.backward(x.double()), f64._forward(zin.double())
        ladj = 0.0
        for l in f64.layers:
            ladj = ladj + l.log_abs_det_jacobian(None, None)
        out["total_ladj64"] = torch.as_tensor(ladj, dtype=torch.float64)
        torch.set_default_dtype(torch.float32)
    arrays = {"x": x.numpy(), "zin": zin.numpy()}
    arrays.update({k: v.detach().numpy() for k, v in out.items()})
    arrays.update({"sd/" + k: v.numpy() for k, v in sd.items()})
    ca = dict(cond_args)
    spec = dict(in_dims=list(in_dims), coupling_blocks=K, cond_args=ca, householder=hh, affine_conjugation=conj,
                masktype=masktype)
    if synth:
        spec["synth_seed"] = seed
    arrays["spec"] = np.array(json.dumps(spec))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    rel = (out["log_prob32"].double() - out["log_prob64"]).abs() / out["log_prob64"].abs()
    print(f"{name:40s} logp[0]={out['log_prob64'][0].item():+.6e} ref32-vs-64 max rel {rel.max().item():.2e} "
          f"max|z|={out['backward64'].abs().max().item():.3g}  {os.path.getsize(path) / 1024:.0f} KB")


def main():
    only = sys.argv[1:]
    # (ConvNet2D's activation is its default nn.ReLU(); padding="same" as in tests/explib/mnist.yaml:62)
    run_case("image_c4_6x6_k3_gated_ln_hh1_conj", (4, 6, 6), 3,
             dict(c_in=4, c_hidden=8, num_layers=2, padding="same", normalize_layers=True, gating=True), 31)
    run_case("image_c16_7x7_k2_plain_channelmask", (16, 7, 7), 2,
             dict(c_in=16, c_hidden=32, num_layers=1, padding="same", normalize_layers=False, gating=False), 32,
             hh=0, conj=False, masktype="channel")
    run_case("image_c3_8x8_k2_gated_hh2_conj", (3, 8, 8), 2,
             dict(c_in=3, c_hidden=6, num_layers=1, padding="same", normalize_layers=False, gating=True), 33, hh=2)
    # the live configurations themselves (tests/explib/mnist.yaml:44-77; experiments/cifar/cifar.yaml:56-77 with 2 of its
    # 10 coupling blocks): the shapes the convolution kernel serves
    run_case("image_mnistcfg_c16_7x7_k2_gated_ln_hh1_conj", (16, 7, 7), 2,
             dict(c_in=16, c_hidden=32, num_layers=1, padding="same", kernel_size=3, normalize_layers=True, gating=True), 34)
    run_case("image_cifarcfg_c48_8x8_k2_gated_ln_hh1_conj", (48, 8, 8), 2,
             dict(c_in=48, c_hidden=32, num_layers=3, padding="same", kernel_size=3, normalize_layers=True, gating=True), 35,
             n=6)
    # the CIFAR configuration in full: all 10 coupling blocks, householder 0 (experiments/cifar/cifar.yaml:56-77);
    # parameters from tests/image_synth.py, so the fixture holds inputs and outputs only
    run_case("image_cifarcfg_full_c48_8x8_k10_gated_ln_hh0_conj_synth", (48, 8, 8), 10,
             dict(c_in=48, c_hidden=32, num_layers=3, padding="same", kernel_size=3, normalize_layers=True, gating=True), 36,
             hh=0, n=6, synth=True)
    # the MNIST configuration with synthesized parameters at 24 rows: the rows the full-batch tests place at the head,
    # middle and tail of 65 536
    run_case("image_mnistcfg_c16_7x7_k2_gated_ln_hh1_conj_synth", (16, 7, 7), 2,
             dict(c_in=16, c_hidden=32, num_layers=1, padding="same", kernel_size=3, normalize_layers=True, gating=True), 37,
             n=24, synth=True)


if __name__ == "__main__":
    main()
