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
