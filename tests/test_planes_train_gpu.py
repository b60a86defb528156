"""Training on the planes pipeline, kernel level, on a real MI355X through the C ABI:
usf_wgrad_blocked_f32 (the weight gradient of F.linear with both operands in the blocked planes format) against fp64
matrix products, and usf_coupling_planes with hidden_out / USF_ACT_GATE (the conditioner's forward with saved activations
and its data-gradient chain) against fp64 torch autograd of the same MLP (networks.py:739-751, transforms.py:277-306)."""
import pytest
import torch

import emulator

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ext():
    from usflows_amd import _ext
    _ext.load()
    return _ext


def _view(buf, M, nkb):
    return buf.cpu().view(torch.bfloat16).view(-(-M // 16), nkb, 3, 64, 8)


def _decode(buf, M, nkb):
    return emulator.planes_decode(_view(buf, M, nkb), M)


def _pack(ext, X, nkb):
    """planes buffer of X [M, C <= 32 nkb] (logical position c = column c)"""
    M, Cn = X.shape
    idx = torch.full((32 * nkb,), -1, dtype=torch.int32)
    idx[:Cn] = torch.arange(Cn, dtype=torch.int32)
    buf = torch.zeros(ext.planes_bytes(M, nkb), dtype=torch.uint8, device=DEV)
    ext.pack_planes(X.to(DEV).contiguous(), buf, M=M, nkb=nkb, idx=idx.to(DEV))
    return buf


WG_CASES = [
    # M, (y_nkb, y_kb0, N), (a_nkb, a_kb0, K), colsum
    (4096, (25, 0, 784), (25, 0, 784), True),        # affine layer: 6 tiles + the folded 16 columns, both ways
    (8229, (25, 12, 400), (8, 0, 256), True),        # conditioner output layer: Y = gradient at the second segment's blocks
    (8192, (25, 0, 392), (8, 0, 256), True),         # ... at the first segment's
    (5000, (8, 0, 256), (8, 0, 256), True),          # hidden layer
    (8192, (8, 0, 256), (25, 12, 400), True),        # first layer, A = conditioning half (second segment)
    (3000, (8, 0, 256), (25, 0, 392), False),
    (100, (3, 1, 50), (4, 0, 100), False),           # small: the plain grid, edge tiles narrower than a block
    (777, (2, 0, 33), (1, 0, 7), False),
    (65536, (25, 0, 784), (25, 0, 784), True),       # BASELINE cfg2 at full size
]


@pytest.mark.parametrize("M,ydesc,adesc,cs", WG_CASES)
def test_wgrad_blocked_matches_fp64(M, ydesc, adesc, cs):
    ext = _ext()
    (y_nkb, y_kb0, N), (a_nkb, a_kb0, K) = ydesc, adesc
    g = torch.Generator().manual_seed(M + N + K)
    # full-width operands: the columns outside the requested ranges must not leak into the result
    Y = torch.randn(M, 32 * y_nkb, generator=g)
    A = torch.randn(M, 32 * a_nkb, generator=g) * 3 + 0.5
    Yb, Ab = _pack(ext, Y, y_nkb), _pack(ext, A, a_nkb)
    G = torch.full((N + 2, K + 3), 7.0, device=DEV)
    csum = torch.full((N,), 5.0, device=DEV) if cs else None
    ext.wgrad_blocked(Yb, y_nkb, y_kb0, Ab, a_nkb, a_kb0, G, M=M, N=N, K=K, ldg=K + 3, alpha=-1.0, beta=0.5,
                      colsum=csum, cs_alpha=2.0, cs_beta=1.0)
    torch.cuda.synchronize()
    Ys, As = Y[:, 32 * y_kb0: 32 * y_kb0 + N].double(), A[:, 32 * a_kb0: 32 * a_kb0 + K].double()
    ref = -(Ys.t() @ As) + 0.5 * 7.0
    mag = Ys.abs().t() @ As.abs()
    got = G.cpu().double()
    assert ((got[:N, :K] - ref).abs() <= 4e-7 * mag + 1e-5).all(), float(((got[:N, :K] - ref).abs() / (mag + 1e-3)).max())
    assert (got[N:] == 7.0).all() and (got[:, K:] == 7.0).all()          # nothing written outside [N, K]
    if cs:
        cref = 2.0 * Ys.sum(0) + 5.0
        assert ((csum.cpu().double() - cref).abs() <= 4e-7 * Ys.abs().sum(0) + 1e-5).all()


def test_wgrad_blocked_is_reproducible_and_ignores_padding_rows_of_A():
    """bitwise the same result launch after launch; rows [M, 16 ceil(M/16)) of A may hold anything finite (Y's are zero)"""
    ext = _ext()
    M, nkb = 8200, 8
    g = torch.Generator().manual_seed(3)
    Y, A = torch.randn(M, 256, generator=g), torch.randn(M, 256, generator=g)
    Yb, Ab = _pack(ext, Y, nkb), _pack(ext, A, nkb)
    G1, G2, G3 = (torch.empty(256, 256, device=DEV) for _ in range(3))
    ext.wgrad_blocked(Yb, nkb, 0, Ab, nkb, 0, G1, M=M, N=256, K=256, ldg=256)
    ext.wgrad_blocked(Yb, nkb, 0, Ab, nkb, 0, G2, M=M, N=256, K=256, ldg=256)
    Apad = torch.randn(-(-M // 16) * 16, 256, generator=g)
    Apad[:M] = A
    Ab2 = _pack(ext, Apad, nkb)                        # M rounded up: the last panel's padding rows now hold values
    ext.wgrad_blocked(Yb, nkb, 0, Ab2, nkb, 0, G3, M=M, N=256, K=256, ldg=256)
    torch.cuda.synchronize()
    assert torch.equal(G1, G2) and torch.equal(G1, G3)


def _mlp_setup(ext, M, nkb, kb_p0, nk_p, kb_t0, nk_t, pass_pos, tr_pos, h, nh, seed):
    """random conditioner + z planes; returns torch fp64 pieces and the device weight images (planes pipeline contract)"""
    from test_planes_gpu import _weight_planes
    g = torch.Generator().manual_seed(seed)
    Z = torch.randn(M, 32 * nkb, generator=g)
    W_in = torch.zeros(h, 32 * nk_p)
    W_in[:, pass_pos - 32 * kb_p0] = torch.randn(h, len(pass_pos), generator=g) / len(pass_pos) ** 0.5
    W_hid = [torch.randn(h, h, generator=g) / h ** 0.5 for _ in range(nh - 1)]
    W_out = torch.zeros(32 * nk_t, h)
    W_out[tr_pos - 32 * kb_t0] = torch.randn(len(tr_pos), h, generator=g) / h ** 0.5
    b_in = torch.randn(h, generator=g) * 0.1
    b_hid = [torch.randn(h, generator=g) * 0.1 for _ in range(nh - 1)]
    b_out = torch.zeros(32 * nk_t)
    b_out[tr_pos - 32 * kb_t0] = torch.randn(len(tr_pos), generator=g) * 0.1
    pad = lambda v, n: torch.cat([v, torch.zeros(n - v.numel())])      # noqa: E731
    dev = dict(W_in=_weight_planes(W_in, 256).to(DEV), b_in=pad(b_in, 256).to(DEV),
               W_hid=[_weight_planes(torch.nn.functional.pad(W, (0, 256 - h)), 256).to(DEV) for W in W_hid],
               b_hid=[pad(b, 256).to(DEV) for b in b_hid],
               W_out=_weight_planes(torch.nn.functional.pad(W_out, (0, 256 - h)), 32 * nk_t).to(DEV), b_out=b_out.to(DEV))
    return Z, (W_in, b_in, W_hid, b_hid, W_out, b_out), dev


def _coupling_desc(ext, z, nkb, M, kb_p0, nk_p, kb_t0, nk_t, nh, dev, sign, slope, act, hidden_out=None, gate=None):
    import ctypes as C
    d = ext.CouplingPlanesDesc()
    d.z, d.z_nkb, d.M = z.data_ptr(), nkb, M
    d.kb_p0, d.nk_p, d.kb_t0, d.nk_t = kb_p0, nk_p, kb_t0, nk_t
    d.n_hidden, d.hidden_padded = nh, 256
    Wi = dev["W_in"]
    d.W_in, d.ldw_in, d.w_in_plane, d.b_in = Wi.data_ptr(), Wi.shape[2], Wi.shape[1] * Wi.shape[2], dev["b_in"].data_ptr()
    for j, (W, b) in enumerate(zip(dev["W_hid"], dev["b_hid"])):
        d.W_hid[j], d.b_hid[j] = W.data_ptr(), b.data_ptr()
        d.ldw_hid, d.w_hid_plane = W.shape[2], W.shape[1] * W.shape[2]
    Wo = dev["W_out"]
    d.W_out, d.ldw_out, d.w_out_plane, d.b_out = Wo.data_ptr(), Wo.shape[2], Wo.shape[1] * Wo.shape[2], dev["b_out"].data_ptr()
    d.sign, d.slope, d.act, d.format = sign, slope, act, 0
    for j in range(nh):
        if hidden_out is not None:
            d.hidden_out[j] = hidden_out[j].data_ptr()
        if gate is not None:
            d.gate[j] = gate[j].data_ptr()
    ext.check(ext.load().usf_coupling_planes(C.byref(d), ext.current_stream(torch.device(DEV))), "usf_coupling_planes")


@pytest.mark.parametrize("nh,h", [(2, 256), (1, 200), (2, 64)])
@pytest.mark.parametrize("M", [16, 1000, 16389])
def test_coupling_planes_saves_its_hidden_activations(M, nh, h):
    """MODE 1: z comes out bit-identical to the inference launch; hidden_out[l] = the layer's activations, exactly the
    values the next layer multiplies (three-way split of the fp32 accumulators)"""
    ext = _ext()
    nkb, kb_p0, nk_p, kb_t0, nk_t = 7, 3, 4, 0, 4           # segments [0, 100) transformed, [100, 200) conditioning
    tr_pos, pass_pos = torch.arange(0, 100), torch.arange(100, 200)
    Z, (W_in, b_in, W_hid, b_hid, W_out, b_out), dev = _mlp_setup(ext, M, nkb, kb_p0, nk_p, kb_t0, nk_t, pass_pos, tr_pos, h, nh, 11)
    z0 = _pack(ext, Z, nkb)
    z1 = z0.clone()
    hout = [torch.zeros(ext.planes_bytes(M, 8), dtype=torch.uint8, device=DEV) for _ in range(nh)]
    _coupling_desc(ext, z0, nkb, M, kb_p0, nk_p, kb_t0, nk_t, nh, dev, -1.0, 0.01, ext.ACT_LEAKY_RELU)
    _coupling_desc(ext, z1, nkb, M, kb_p0, nk_p, kb_t0, nk_t, nh, dev, -1.0, 0.01, ext.ACT_LEAKY_RELU, hidden_out=hout)
    torch.cuda.synchronize()
    assert torch.equal(z0, z1)
    act = lambda v: torch.where(v > 0, v, v * 0.01)      # noqa: E731
    a = act(Z[:, 32 * kb_p0: 32 * (kb_p0 + nk_p)].double() @ W_in.double().t() + b_in.double())
    refs = [a]
    for W, b in zip(W_hid, b_hid):
        a = act(a @ W.double().t() + b.double())
        refs.append(a)
    for l in range(nh):
        got = _decode(hout[l], M, 8).double()
        assert (got[:, h:] == 0).all()
        assert ((got[:, :h] - refs[l]).abs().max() / refs[l].abs().max()).item() < 2e-6, l
    out = Z.double().clone()
    out[:, tr_pos] -= (refs[-1] @ W_out.double().t() + b_out.double())[:, tr_pos - 32 * kb_t0]
    got = _decode(z1, M, nkb).double()
    assert ((got - out).abs().max() / out.abs().max()).item() < 2e-6


@pytest.mark.parametrize("nh,h,slope", [(2, 256, 0.01), (1, 256, 0.2), (2, 96, 0.0)])
@pytest.mark.parametrize("M", [16, 1000, 16389])
def test_coupling_planes_gate_mode_is_the_conditioners_backward(M, nh, h, slope):
    """MODE 2 on the gradient buffer: g[:, pass] += sign * d MLP(z_pass)^T g[:, trans] and hidden_out[l] = the gradients at
    the pre-activations, against fp64 torch autograd of the MLP"""
    ext = _ext()
    from test_planes_gpu import _weight_planes
    nkb, kb_p0, nk_p, kb_t0, nk_t = 7, 3, 4, 0, 4
    tr_pos, pass_pos = torch.arange(0, 100), torch.arange(100, 200)
    Z, (W_in, b_in, W_hid, b_hid, W_out, b_out), dev = _mlp_setup(ext, M, nkb, kb_p0, nk_p, kb_t0, nk_t, pass_pos, tr_pos, h, nh, 23)
    sign = -1.0
    # forward with saved activations
    z = _pack(ext, Z, nkb)
    hsave = [torch.zeros(ext.planes_bytes(M, 8), dtype=torch.uint8, device=DEV) for _ in range(nh)]
    _coupling_desc(ext, z, nkb, M, kb_p0, nk_p, kb_t0, nk_t, nh, dev, sign, slope, ext.ACT_LEAKY_RELU, hidden_out=hsave)
    # fp64 chain rule of out = z; out_T += sign * MLP(z_P), with the gates the DEVICE saved (a hidden unit whose
    # pre-activation is within fp32 noise of zero may sit on the other branch than in fp64: checked separately below)
    torch.cuda.synchronize()
    hdev = [_decode(b, M, 8).double()[:, :h] for b in hsave]
    act = lambda v: torch.where(v > 0, v, v * slope)      # noqa: E731
    a = act(Z[:, 32 * kb_p0: 32 * (kb_p0 + nk_p)].double() @ W_in.double().t() + b_in.double())
    h64 = [a]
    for W, b in zip(W_hid, b_hid):
        a = act(a @ W.double().t() + b.double())
        h64.append(a)
    for l in range(nh):
        flips = ((hdev[l] > 0) != (h64[l] > 0)).double().mean().item()
        assert flips < 1e-5, (l, flips)
    gen = torch.Generator().manual_seed(5)
    Gout = torch.randn(M, 32 * nkb, generator=gen)
    gate = lambda v, hh: torch.where(hh > 0, v, v * slope)      # noqa: E731
    v = Gout[:, 32 * kb_t0: 32 * (kb_t0 + nk_t)].double() @ W_out.double()
    d_refs = {nh - 1: gate(v, hdev[nh - 1])}
    for l in range(nh - 2, -1, -1):
        d_refs[l] = gate(d_refs[l + 1] @ W_hid[l].double(), hdev[l])
    ref = Gout.double().clone()
    ref[:, 32 * kb_p0: 32 * (kb_p0 + nk_p)] += sign * (d_refs[0] @ W_in.double())
    # device: transposed images, roles of the block ranges swapped, zero biases
    gbuf = _pack(ext, Gout, nkb)
    zeros = torch.zeros(max(256, 32 * nk_p), device=DEV)
    Wo_t = torch.zeros(h, 32 * nk_t)
    Wo_t[:, tr_pos - 32 * kb_t0] = W_out.t()[:, tr_pos - 32 * kb_t0]
    Wi_t = torch.zeros(32 * nk_p, h)
    Wi_t[pass_pos - 32 * kb_p0] = W_in.t()[pass_pos - 32 * kb_p0]
    bdev = dict(W_in=_weight_planes(Wo_t, 256).to(DEV), b_in=zeros,
                W_hid=[_weight_planes(torch.nn.functional.pad(W.t(), (0, 256 - h)), 256).to(DEV) for W in reversed(W_hid)],
                b_hid=[zeros for _ in W_hid],
                W_out=_weight_planes(torch.nn.functional.pad(Wi_t, (0, 256 - h)), 32 * nk_p).to(DEV), b_out=zeros)
    dh = [torch.zeros(ext.planes_bytes(M, 8), dtype=torch.uint8, device=DEV) for _ in range(nh)]
    _coupling_desc(ext, gbuf, nkb, M, kb_t0, nk_t, kb_p0, nk_p, nh, bdev, sign, slope, ext.ACT_GATE,
                   hidden_out=dh, gate=list(reversed(hsave)))
    torch.cuda.synchronize()
    got = _decode(gbuf, M, nkb).double()
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 3e-6
    for l in range(nh):                                   # dh[l] = gradient at the pre-activation of forward layer nh - 1 - l
        d_ref = d_refs[nh - 1 - l]                        # (the kernel applies the sign where the result leaves the MLP)
        d_got = _decode(dh[l], M, 8).double()[:, :h]
        assert ((d_got - d_ref).abs().max() / d_ref.abs().max()).item() < 3e-6, l
