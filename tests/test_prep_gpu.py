"""Parameter-prep kernels (usf_prep.hip) through the C ABI against the oracle's fp64
restatement of LUTransform / HouseholderTransform / SequentialAffineTransform (transforms.py:1271-1320,
795-809, 1457-1476).  fp64 on both sides: tolerance 1e-11 relative to the matrix scale; the
fp32 packing (usf_pack_weight_f32) is bit-exact against its documented semantics (tests/emulator.py)."""
import pytest
import torch

import emulator
from oracle import usflows_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ext():
    from usflows_amd import _ext
    _ext.load()
    return _ext


def _lu_params(D, n, seed, alpha=0.3):
    g = torch.Generator().manual_seed(seed)
    Ls = [(alpha * torch.randn(D, D, generator=g)).contiguous() for _ in range(n)]
    Us = []
    for _ in range(n):
        U = alpha * torch.randn(D, D, generator=g)
        d = (0.75 + 0.5 * torch.rand(D, generator=g)) * torch.where(torch.rand(D, generator=g) < 0.5, -1.0, 1.0)
        U = U.triu(1) + torch.diag(d) + U.tril(-1)       # garbage below the diagonal must be ignored
        Us.append(U.contiguous())
    return Ls, Us


@pytest.mark.parametrize("D,n", [(1, 1), (5, 2), (31, 1), (32, 3), (33, 2), (64, 1), (100, 3), (257, 2), (784, 3)])
def test_lu_prepare_matches_oracle(D, n):
    ext = _ext()
    Ls, Us = _lu_params(D, n, seed=D * 7 + n, alpha=min(0.3, 0.8 / D ** 0.5))   # keeps cond(L), cond(U) moderate
    out = ext.lu_prepare([t.to(DEV) for t in Ls], [t.to(DEV) for t in Us], keep_factors=True)
    torch.cuda.synchronize()
    for i in range(n):
        L64, U64 = Ls[i].double(), Us[i].double()
        M_ref = orc.lu_matrix(L64, U64)
        Minv_ref = orc.lu_inverse_matrix(L64, U64)
        scale_m, scale_i = M_ref.abs().max().item(), Minv_ref.abs().max().item()
        assert (out["M"][i].cpu() - M_ref).abs().max().item() <= 1e-12 * max(1.0, scale_m)
        assert (out["Minv"][i].cpu() - Minv_ref).abs().max().item() <= 1e-10 * max(1.0, scale_i)
        assert abs(out["ladj"][i].item() - orc.lu_ladj(U64).item()) <= 1e-11 * max(1.0, D)
        assert torch.equal(out["tri"][2 * i].cpu(), orc.lu_L(L64))
        assert torch.equal(out["tri"][2 * i + 1].cpu(), orc.lu_U(U64).t())
        Linv_ref = torch.linalg.solve_triangular(orc.lu_L(L64), torch.eye(D, dtype=torch.float64), upper=False)
        assert (out["tri_inv"][2 * i].cpu() - Linv_ref).abs().max().item() <= 1e-10 * max(1.0, Linv_ref.abs().max().item())
        # the inverse of a lower-triangular matrix is lower-triangular: exact zeros above the diagonal
        assert out["tri_inv"][2 * i].cpu().triu(1).abs().max().item() == 0.0
        assert out["tri_inv"][2 * i + 1].cpu().triu(1).abs().max().item() == 0.0
        # known-answer property of the reference's own test (tests/veriflow/transforms_test.py:35-51): M Minv = I
        eye_err = (out["M"][i] @ out["Minv"][i] - torch.eye(D, dtype=torch.float64, device=DEV)).abs().max().item()
        assert eye_err <= 1e-9 * max(1.0, scale_m * scale_i)


def test_lu_prepare_identity_known_answer():
    """LU init of the reference's transforms_test.py:35-51: L_raw = 0, U_raw = I -> M = M^-1 = I, ladj = 0"""
    ext = _ext()
    D = 48
    out = ext.lu_prepare([torch.zeros(D, D, device=DEV)], [torch.eye(D, device=DEV)])
    eye = torch.eye(D, dtype=torch.float64, device=DEV)
    assert torch.equal(out["M"][0], eye) and torch.equal(out["Minv"][0], eye)
    assert out["ladj"][0].item() == 0.0


def test_lu_prepare_reference_known_answer():
    """the reference's own LU test (tests/veriflow/transforms_test.py:35-51): L = tril(ones), U = I, b = 0 =>
    y = M x = arange(dim) + 1 for x = ones, exact inverse, log|det J| = 0 -- through the device prep kernels"""
    ext = _ext()
    D = 10
    out = ext.lu_prepare([torch.ones(D, D, device=DEV)], [torch.eye(D, device=DEV)])
    ones = torch.ones(D, dtype=torch.float64, device=DEV)
    y = out["M"][0] @ ones
    assert torch.equal(y.cpu(), torch.arange(D, dtype=torch.float64) + 1)
    assert torch.equal((out["Minv"][0] @ y).cpu(), torch.ones(D, dtype=torch.float64))
    assert out["ladj"][0].item() == 0.0


@pytest.mark.parametrize("transA", [False, True])
@pytest.mark.parametrize("transB", [False, True])
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (7, 5, 3), (64, 64, 16), (65, 130, 33), (200, 77, 129)])
def test_gemm_f64(transA, transB, M, N, K):
    ext = _ext()
    g = torch.Generator().manual_seed(M * 1000 + N * 10 + K)
    A = torch.randn((K, M) if transA else (M, K), generator=g, dtype=torch.float64)
    B = torch.randn((N, K) if transB else (K, N), generator=g, dtype=torch.float64)
    C0 = torch.randn(3, M, N, generator=g, dtype=torch.float64)
    ref = 0.5 * ((A.t() if transA else A) @ (B.t() if transB else B)) - 2.0 * C0
    Ad, Bd, Cd = A.to(DEV), B.to(DEV), C0.to(DEV)
    # batch of 3 sharing A and B (stride 0), distinct C
    ext.gemm_f64(Ad, Bd, Cd, M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], ldc=N, transA=transA, transB=transB,
                 batch=3, strideC=M * N, alpha=0.5, beta=-2.0)
    torch.cuda.synchronize()
    assert (Cd.cpu() - ref).abs().max().item() <= 1e-12 * max(1.0, K)
    got = ext.matmul_f64(Ad, Bd, transA, transB).cpu()
    assert (got - (A.t() if transA else A) @ (B.t() if transB else B)).abs().max().item() <= 1e-12 * max(1.0, K)


@pytest.mark.parametrize("D", [40, 130])
def test_gemm_f64_triangular_hints(D):
    ext = _ext()
    g = torch.Generator().manual_seed(D)
    Lo = torch.randn(D, D, generator=g, dtype=torch.float64).tril()
    Up = torch.randn(D, D, generator=g, dtype=torch.float64).triu()
    assert torch.equal(ext.matmul_f64(Lo.to(DEV), Up.to(DEV), tri=1), ext.matmul_f64(Lo.to(DEV), Up.to(DEV)))
    assert torch.equal(ext.matmul_f64(Up.to(DEV), Lo.to(DEV), tri=2), ext.matmul_f64(Up.to(DEV), Lo.to(DEV)))


@pytest.mark.parametrize("D", [70, 200])
def test_gemm_f64_one_sided_hints_and_output_masks(D):
    """k <= i (3) / k <= j (4) hints, and the masks that compute only the tiles touching one triangle (+8 upper,
    +16 lower; the other tiles of C are left untouched): what the batched LU chain rule of training.py uses"""
    ext = _ext()
    g = torch.Generator().manual_seed(D)
    X = torch.randn(D, D, generator=g, dtype=torch.float64)
    Lo = torch.randn(D, D, generator=g, dtype=torch.float64).tril()
    full = lambda A, B, **kw: ext.matmul_f64(A.to(DEV), B.to(DEV), **kw).cpu()
    assert torch.equal(full(Lo, X, tri=3), full(Lo, X))                      # op(A) lower-triangular
    assert torch.equal(full(X, Lo.t().contiguous(), tri=4), full(X, Lo.t().contiguous()))   # op(B) upper-triangular
    for mask, keep in ((8, torch.triu), (16, torch.tril)):
        C = torch.full((D, D), 7.0, dtype=torch.float64, device=DEV)
        ext.gemm_f64(X.to(DEV), X.to(DEV), C, M=D, N=D, K=D, lda=D, ldb=D, ldc=D, tri=mask)
        ref = X @ X
        got = C.cpu()
        assert (keep(got) - keep(ref)).abs().max().item() <= 1e-11 * D         # the wanted triangle is complete
        r, c = torch.meshgrid(torch.arange(D), torch.arange(D), indexing="ij")
        far = (r // 64 > c // 64) if mask == 8 else (c // 64 > r // 64)        # tiles strictly on the other side
        assert (got[far] == 7.0).all()


@pytest.mark.parametrize("D,nvs", [(7, 1), (64, 2), (100, 3), (784, 2)])
def test_householder_matches_oracle(D, nvs):
    ext = _ext()
    g = torch.Generator().manual_seed(D + nvs)
    w0 = torch.zeros(D, D)
    w0[torch.arange(D), torch.randperm(D, generator=g)] = 1.0
    vk = 0.2 * torch.randn(nvs, D, generator=g)
    got = ext.householder(w0.to(DEV), vk.to(DEV)).cpu()
    ref = orc.householder_matrix(vk.double(), w0.double())
    assert (got - ref).abs().max().item() <= 1e-13
    assert (got @ got.t() - torch.eye(D, dtype=torch.float64)).abs().max().item() <= 1e-12     # orthogonal


@pytest.mark.parametrize("src_dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("transpose", [False, True])
def test_pack_weight_bit_exact(src_dtype, transpose):
    ext = _ext()
    g = torch.Generator().manual_seed(3)
    R, Cc = 37, 53
    src = torch.randn(R, Cc, generator=g, dtype=torch.float64).to(src_dtype)
    n_out, n_in, ldw, ldp = 44, 40, 48, 64
    rows_src = Cc if transpose else R
    cols_src = R if transpose else Cc
    oi = torch.randint(-1, rows_src, (n_out,), generator=g).to(torch.int32)
    ii = torch.randint(-1, cols_src, (n_in,), generator=g).to(torch.int32)
    W = torch.full((n_out, ldw), 7.0)
    planes = torch.ones(3, n_out, ldp, dtype=torch.bfloat16)
    W_ref, planes_ref = W.clone(), planes.clone()
    emulator._emu_pack_weight(src, oi, n_out, ii, n_in, W=W_ref, ldw=ldw, planes=planes_ref, transpose=transpose)
    Wd, Pd = W.to(DEV), planes.to(DEV)
    ext.pack_weight(src.to(DEV), oi.to(DEV), n_out, ii.to(DEV), n_in, W=Wd, ldw=ldw, planes=Pd, transpose=transpose)
    torch.cuda.synchronize()
    assert torch.equal(Wd.cpu(), W_ref)                       # columns >= n_in of W untouched (7.0)
    assert torch.equal(Pd.cpu().view(torch.int16), planes_ref.view(torch.int16))
    # the three planes sum back to the fp32 value to the last bit (8+8+8 significant bits)
    s = Pd[0].float() + Pd[1].float() + Pd[2].float()
    assert torch.equal(s[:, :n_in].cpu(), W_ref[:, :n_in])


@pytest.mark.parametrize("planes_dtype", [None, torch.bfloat16, torch.float16])
def test_batched_pack_jobs_transposed_through_tiles_bit_exact(planes_dtype):
    """a batch of pack jobs (``batch_jobs``): the transposed ones leave through usf_pack_weights_t_f32 (32 x 32 LDS tiles),
    the others through usf_pack_weights_f32 -- every image bit for bit the documented gather (emulator), with index
    selections, holes (-1), ragged shapes that end inside a tile, fp32 and fp64 sources, W and / or planes"""
    ext = _ext()
    g = torch.Generator().manual_seed(11)
    jobs = []
    # (the last field: ldw - n_in; the non-transposed jobs with 16-byte-aligned image rows take usf_pack_weights_f32's 4-column form)
    for k, (R, Cc, n_out, n_in, tr, dt, pad) in enumerate([(70, 91, 95, 64, True, torch.float64, 3), (33, 33, 33, 33, True, torch.float32, 3),
                                                           (784, 784, 784, 800, True, torch.float64, 3), (40, 57, 31, 40, False, torch.float32, 3),
                                                           (5, 130, 129, 4, True, torch.float32, 3), (64, 64, 64, 64, True, torch.float64, 3),
                                                           (784, 784, 784, 784, False, torch.float64, 0), (100, 300, 77, 260, False, torch.float32, 0),
                                                           (20, 2100, 9, 2100, False, torch.float64, 4), (50, 61, 50, 61, False, torch.float64, 3),
                                                           (128, 96, 128, 90, False, torch.float32, 2)]):
        src = torch.randn(R, Cc, generator=g, dtype=torch.float64).to(dt)
        rows_src, cols_src = (Cc, R) if tr else (R, Cc)
        oi = torch.randint(-1, rows_src, (n_out,), generator=g).to(torch.int32) if k % 2 == 0 else None
        ii = torch.randint(-1, cols_src, (n_in,), generator=g).to(torch.int32) if k % 3 != 1 else None
        if oi is None:
            n_out = min(n_out, rows_src)
        if ii is None:
            n_in = min(n_in, cols_src)
        ldw, ldp = n_in + pad, (n_in + 31) // 32 * 32
        W = torch.full((n_out, ldw), 7.0)
        planes = None
        if planes_dtype is not None:
            planes = torch.ones(2 if planes_dtype == torch.float16 else 3, n_out, ldp, dtype=planes_dtype)
        jobs.append(dict(src=src, oi=oi, n_out=n_out, ii=ii, n_in=n_in, W=W, ldw=ldw, planes=planes, tr=tr))
    dev = lambda t: None if t is None else t.to(DEV)
    got = []
    with ext.batch_jobs(torch.device(DEV)) as bj:
        for j in jobs:
            Wd, Pd = dev(j["W"]), dev(j["planes"])
            ext.pack_weight(dev(j["src"]), dev(j["oi"]), j["n_out"], dev(j["ii"]), j["n_in"], W=Wd, ldw=j["ldw"], planes=Pd,
                            transpose=j["tr"])
            got.append((Wd, Pd))
        assert len(bj.jobs) == len(jobs)
    torch.cuda.synchronize()
    for j, (Wd, Pd) in zip(jobs, got):
        W_ref = j["W"].clone()
        P_ref = None if j["planes"] is None else j["planes"].clone()
        emulator._emu_pack_weight_now(j["src"], j["oi"], j["n_out"], j["ii"], j["n_in"], W=W_ref, ldw=j["ldw"], planes=P_ref,
                                      transpose=j["tr"])
        assert torch.equal(Wd.cpu(), W_ref), (j["n_out"], j["n_in"], j["tr"])
        if P_ref is not None:
            assert torch.equal(Pd.cpu().view(torch.int16), P_ref.view(torch.int16)), (j["n_out"], j["n_in"], j["tr"])


def test_matvec_f64():
    ext = _ext()
    g = torch.Generator().manual_seed(5)
    A = torch.randn(70, 91, generator=g, dtype=torch.float64)
    b = torch.randn(91, generator=g, dtype=torch.float64)
    idx = torch.tensor([3, -1, 69, 0, 0, 12, -1], dtype=torch.int32)
    o32 = torch.empty(7, device=DEV)
    o64 = torch.empty(7, dtype=torch.float64, device=DEV)
    ext.matvec_f64(A.to(DEV), b.to(DEV), idx=idx.to(DEV), n_out=7, alpha=-1.0, out32=o32, out64=o64)
    ref = torch.zeros(7, dtype=torch.float64)
    ok = idx >= 0
    ref[ok] = -(A[idx[ok].long()] @ b)
    assert (o64.cpu() - ref).abs().max().item() <= 1e-13
    assert torch.equal(o32.cpu(), o64.cpu().float())


def _ref_affine(t):
    """(M, Minv, b) in fp64 from a block's parameters with the oracle's statements of the reference formulas"""
    from usflows_amd import transforms as T
    d64 = lambda p: p.detach().cpu().double()
    if isinstance(t, T.LUTransform):
        return orc.lu_matrix(d64(t.L_raw), d64(t.U_raw)), orc.lu_inverse_matrix(d64(t.L_raw), d64(t.U_raw)), d64(t.bias_vector)
    if isinstance(t, T.HouseholderTransform):
        M = orc.householder_matrix(d64(t.vk_householder), d64(t.w_0))
        return M, M.t().contiguous(), torch.zeros(t.dim, dtype=torch.float64)
    parts = [_ref_affine(u) for u in t.transforms]                 # transforms.py:1457-1476
    M = torch.eye(t.dim, dtype=torch.float64)
    Minv = torch.eye(t.dim, dtype=torch.float64)
    b = torch.zeros(t.dim, dtype=torch.float64)
    for m, _, bi in parts:
        M = M @ m
        b = b @ m + bi
    for _, mi, _ in parts[::-1]:
        Minv = Minv @ mi
    return M, Minv, b


@pytest.mark.parametrize("case", ["synth_d64_k4_hh1_conj_laplace", "synth_d33_k3_lu2_hh1", "synth_d16_k4_hh2_conj_laplace"])
def test_engine_prep_matches_oracle(case):
    """the engine's prep (all affine blocks through ONE batched prepare + Sequential composition on the f64 GEMM)
    against the oracle's matrices, for the [LU.., Householder] compositions the reference's live configs use"""
    from golden_util import load_case
    from model_util import build_flow
    from usflows_amd.engine import FlowEngine, prepare_affine_blocks
    spec, sd, a = load_case(case)
    flow = build_flow(spec, sd, device=DEV)
    eng = FlowEngine(flow.layers)
    blocks = list({id(s.module): s.module for s in eng.steps if s.kind == "affine"}.values())
    res = prepare_affine_blocks(blocks, DEV)
    torch.cuda.synchronize()
    for blk in blocks:
        M_ref, Minv_ref, b_ref = _ref_affine(blk)
        r = res[id(blk)]
        assert (r["M"].cpu() - M_ref).abs().max().item() <= 1e-11 * max(1.0, M_ref.abs().max().item())
        assert (r["Minv"].cpu() - Minv_ref).abs().max().item() <= 1e-9 * max(1.0, Minv_ref.abs().max().item())
        assert (r["b"].cpu() - b_ref).abs().max().item() <= 1e-12 * max(1.0, b_ref.abs().max().item())


@pytest.mark.parametrize("with_m", [False, True])
@pytest.mark.parametrize("n,D", [(1, 5), (3, 33), (4, 784)])
def test_lu_grad_finish_is_the_torch_formulation(n, D, with_m):
    """usf_lu_grad_finish_f64 == tril(dL + TL, -1), triu(dU) + diag(c / U_jj) + triu(TU) rounded once to fp32
    (what Flow.fit's autograd derives through transforms.py:1271-1320), garbage in the unwanted triangles included"""
    ext = _ext()
    g = torch.Generator().manual_seed(n * D)
    r = lambda *shape: torch.randn(*shape, generator=g, dtype=torch.float64)
    dL, dU, TL, TU, c, tri = r(n, D, D), r(n, D, D), r(n, D, D), r(n, D, D), r(n), r(2 * n, D, D)
    dL = dL + torch.full_like(dL, 1e30).triu()                 # never read: the upper triangle (diagonal included) of dL
    dU = dU + torch.full_like(dU, 1e30).tril(-1)               # ... the strictly lower triangle of dU
    oL = torch.full((n * D * D,), 7.0, device=DEV)
    oU = torch.full((n * D * D,), 7.0, device=DEV)
    dev = lambda t: t.to(DEV)
    ext.lu_grad_finish(dev(dL), dev(dU), dev(TL) if with_m else None, dev(TU) if with_m else None, dev(c), dev(tri), n, D, oL, oU)
    torch.cuda.synchronize()
    rl = dL.tril(-1) + (TL.tril(-1) if with_m else 0.0)
    ru = dU.triu() + torch.diag_embed(c[:, None] / tri[1::2].diagonal(dim1=1, dim2=2))
    if with_m:
        ru = ru + TU.triu()
    assert torch.equal(oL.cpu().view(n, D, D), rl.float())
    assert torch.equal(oU.cpu().view(n, D, D), ru.float())
