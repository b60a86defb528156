"""Host-side proof that every usf_linear_f32 instantiation the BASELINE configurations select is one the kernel
parity tests (tests/test_kernels_gpu.py, shapes in tests/shapes.py) run against reference arithmetic.
usf_linear_variant is pure dispatch logic: no GPU needed."""
import ctypes as C

import shapes


def _variant(M, N, K):
    from usflows_amd import _ext
    lib = _ext.load()
    d = _ext.LinearDesc()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldw, d.ldc = K, K, N
    d.A = d.W = d.C = 1 << 20                      # any 16-byte aligned address: nothing is launched
    d.W_split = 1 << 20
    d.ldw_split = (K + 31) // 32 * 32
    d.split_plane_stride = N * d.ldw_split
    return lib.usf_linear_variant(C.byref(d))


def test_variant_codes_of_the_headline_shapes():
    assert _variant(100, 784, 784) == 1000                     # small-batch kernel
    assert _variant(65536, 784, 784) // 1000 == 3              # bf16x3 tiles
    assert _variant(32768, 3072, 3072) // 1000 == 3
    # without planes the same shape takes the exact-f32 tile
    from usflows_amd import _ext
    d = _ext.LinearDesc()
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc = 65536, 784, 784, 784, 784, 784
    d.A = d.W = d.C = 1 << 20
    assert _ext.load().usf_linear_variant(C.byref(d)) == 2254


def test_every_baseline_instantiation_is_parity_tested():
    tested = {_variant(*s) for s in shapes.BF16X3_SMALL + shapes.BF16X3_BIG}
    for cfg, lst in shapes.BASELINE_LINEAR_SHAPES.items():
        for s in lst:
            assert _variant(*s) in tested, (cfg, s, _variant(*s), sorted(tested))
    # and the D x D affine shapes (the dominant kernel) plus all of cfg4's are in the kernel test list at full size
    for cfg, lst in shapes.BASELINE_LINEAR_SHAPES.items():
        for s in lst:
            if s[1] >= 784 or cfg.startswith("cfg4"):
                assert s in shapes.BF16X3_BIG, (cfg, s)


def _planes_variant(M, nk, out):
    from usflows_amd import _ext
    d = _ext.GemmPlanesDesc()
    d.M, d.nk, d.a_nkb = M, nk, nk
    if out < 0:
        d.C_f32, d.N = 1 << 20, -out
    else:
        d.C_planes, d.c_kbn, d.c_nkb = 1 << 20, out, out
    return _ext.load().usf_gemm_planes_variant(C.byref(d))


def test_every_baseline_planes_instantiation_is_parity_tested():
    tested = {_planes_variant(*s) for s in shapes.PLANES_TESTED}
    assert tested == {5040, 5041, 5050, 5051}                  # all four instantiations of gemm_planes_kernel
    for cfg, lst in shapes.BASELINE_PLANES_SHAPES.items():
        for s in lst:
            assert _planes_variant(*s) in tested, (cfg, s)


def test_cfg1_moons_plumbing_on_cpu():
    """BASELINE configs[0] (plumbing, no GPU): the 2-D moons data set (datasets.py:181-183: sklearn make_moons, noise 0.05),
    USFlow(in_dims=[2], 4 additive coupling blocks, ConditionalDenseNN [32, 32] + ReLU, Laplace base), batch 4096 through
    the CPU formulation of the mirror: log_prob / backward / _forward against the oracle, the UDL constant, and a short
    Flow.fit that lowers the loss"""
    import numpy as np
    import torch
    from sklearn.datasets import make_moons
    from oracle import usflows_oracle as orc
    from usflows_amd.synth import build_usflow
    xs, _ = make_moons(n_samples=4096, noise=0.05, random_state=0)
    x = torch.from_numpy(xs.astype(np.float32))
    spec = orc.FlowSpec(2, 4, [32, 32], householder=1, negative_slope=0.0, conditioner="ConditionalDenseNN", base="laplace")
    sd = orc.synth_state_dict(spec, seed=1, alpha=0.3)
    flow = build_usflow(spec, sd, device="cpu")
    with torch.no_grad():
        lp, z = flow.log_prob(x), flow.backward(x)
        xr = flow._forward(z)
    sd64 = orc.to_dtype(sd, torch.float64)
    ref = orc.flow_log_prob(sd64, spec, x.double())
    assert ((lp.double() - ref).abs() / ref.abs()).max().item() < 1e-5
    assert (z.double() - orc.flow_backward(sd64, spec, x.double())).abs().max().item() < 1e-4
    assert (xr - x).abs().max().item() < 1e-4
    const = lp.double() - torch.distributions.Laplace(0.0, 1.0).log_prob(z.double()).sum(-1)
    assert (const + float(orc.total_ladj(sd64, spec))).abs().max().item() < 1e-5      # uniformly scaling: one constant

    class DS:
        def __len__(self):
            return x.shape[0]

        def __getitem__(self, i):
            return (x[i],)

    losses = flow.fit(DS(), torch.optim.Adam, dict(lr=2e-3), batch_size=256, shuffle=False, device=torch.device("cpu"), epochs=3)
    assert losses[-1] < losses[0]
