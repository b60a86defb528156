"""Training of image-shaped flows on the device (SURVEY rows N2 x N4; usflows_amd/image_training.py): the gradient kernels
against fp64 torch autograd of the same expressions, whole-flow gradients and a Flow.fit run against the REAL reference
(tests/golden/imagegrads_*.npz, imagefit_*.npz; made by tests/golden/make_golden_image_grads.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from golden_util import grads_close, image_grad_case_names, load_image_case, load_image_fit, load_image_grads

DEV = "cuda:0"


def _close(got, want, tol=1e-5, what=""):
    want = want.double().cpu()
    got = got.double().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    s = max(want.abs().max().item(), 1e-30)
    err = (got - want).abs().max().item()
    assert err <= tol * s, f"{what}: max abs err {err:.3e} vs scale {s:.3e} (rel {err / s:.2e})"


@pytest.mark.parametrize("name", image_grad_case_names())
def test_mirror_autograd_matches_reference_gradients_cpu(name):
    """the mirror's torch formulation under autograd (fp64, CPU) reproduces the reference's gradients: pins what the device
    path below is compared with"""
    flow, a = load_image_case(name)
    loss_ref, g_ref = load_image_grads(name)
    torch.set_default_dtype(torch.float64)
    try:
        # the same flow built in fp64 (constructor under the fp64 default, parameters copied from the fp32 mirror)
        from usflows_amd.flows import USFlow
        from usflows_amd.networks import ConvNet2D
        import json, os
        from golden_util import GOLDEN_DIR
        d = json.loads(str(np.load(os.path.join(GOLDEN_DIR, name + ".npz"))["spec"]))
        dims = d["in_dims"]
        f64 = USFlow(torch.distributions.Laplace(torch.zeros(dims), torch.ones(dims)), dims, d["coupling_blocks"], ConvNet2D,
                     dict(d["cond_args"]), householder=d["householder"], affine_conjugation=d["affine_conjugation"],
                     masktype=d["masktype"])
        f64.load_state_dict({k: v.double() for k, v in flow.state_dict().items()}, strict=True)
        f64 = f64.double()                                  # (the Householder permutation is created as fp32 on purpose)
        lp = f64.log_prob(a["x"].double())
        loss = -lp.mean()
        loss.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    assert abs(float(loss.detach()) - loss_ref) < 1e-9 * abs(loss_ref)
    named = dict(f64.named_parameters())
    assert set(g_ref) <= set(named)
    for k, g in g_ref.items():
        _close(named[k].grad, g, 1e-8, k)


# ---- kernels ---------------------------------------------------------------------------------------------------------
WG_SHAPES = [  # (cin, cout, H, W, ks)
    (16, 32, 7, 7, 3), (32, 32, 7, 7, 3), (32, 16, 7, 7, 3), (32, 64, 7, 7, 1), (16, 16, 7, 7, 1),
    (48, 32, 8, 8, 3), (32, 32, 8, 8, 3), (32, 48, 8, 8, 3), (32, 64, 8, 8, 1), (48, 48, 8, 8, 1), (64, 32, 8, 8, 1),
    (16, 16, 3, 5, 3), (32, 16, 1, 2, 3), (16, 48, 8, 8, 3), (32, 32, 6, 8, 1), (16, 32, 2, 2, 1),
]


def _wgrad_ref(x, dy, ks, in_mul=None, pre_sub=None, act=None):
    x = x.double()
    if pre_sub is not None:
        x = x - pre_sub.double().view(1, -1, 1, 1)
    if act is not None:
        x = F.leaky_relu(x, act)
    if in_mul is not None:
        x = x * in_mul.double().view(1, *x.shape[1:])
    B, cin, H, W = x.shape
    cols = F.unfold(x, ks, padding=ks // 2)                                   # [B, cin * ks * ks, HW]
    dW = torch.einsum("bop,bkp->ok", dy.double().flatten(2), cols).reshape(dy.shape[1], cin, ks, ks)
    return dW, dy.double().sum(dim=(0, 2, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", WG_SHAPES)
@pytest.mark.parametrize("B", [1, 5, 300])
def test_conv_wgrad_vs_fp64(shape, B):
    from usflows_amd import _ext
    cin, cout, H, W, ks = shape
    g = torch.Generator().manual_seed(cin * 1000 + cout * 10 + ks + B)
    x = torch.randn(B, cin, H, W, generator=g).to(DEV)
    dy = torch.randn(B, cout, H, W, generator=g).to(DEV)
    r = _ext.conv_wgrad(x, dy, ks)
    assert r is not None, "shape not served"
    dW, db = r
    rW, rb = _wgrad_ref(x, dy, ks)
    _close(dW, rW, 2e-6, "dW")
    _close(db, rb, 2e-6, "db")
    # the input transforms of the forward kernels
    in_mul = (torch.rand(cin * H * W, generator=g) < 0.5).float().to(DEV)
    pre = torch.randn(cin, generator=g).to(DEV)
    r = _ext.conv_wgrad(x, dy, ks, in_mul=in_mul, in_act=_ext.ACT_LEAKY_RELU, in_slope=0.0, want_bias=False)
    assert r[1] is None
    _close(r[0], _wgrad_ref(x, dy, ks, in_mul=in_mul, act=0.0)[0], 2e-6, "dW relu mask")
    r = _ext.conv_wgrad(x, dy, ks, pre_sub=pre, in_act=_ext.ACT_LEAKY_RELU, in_slope=0.1)
    _close(r[0], _wgrad_ref(x, dy, ks, pre_sub=pre, act=0.1)[0], 2e-6, "dW pre_sub leaky")
    # deterministic: the same launch twice gives the same bits
    r2 = _ext.conv_wgrad(x, dy, ks, pre_sub=pre, in_act=_ext.ACT_LEAKY_RELU, in_slope=0.1)
    assert torch.equal(r[0], r2[0]) and torch.equal(r[1], r2[1])


@pytest.mark.gpu
def test_conv_wgrad_full_batch_and_unserved_shapes():
    from usflows_amd import _ext
    g = torch.Generator().manual_seed(9)
    B = 65536
    x = torch.randn(B, 32, 7, 7, generator=g).to(DEV)
    dy = torch.randn(B, 32, 7, 7, generator=g).to(DEV)
    dW, db = _ext.conv_wgrad(x, dy, 3)
    ref = torch.zeros(32, 32, 3, 3, dtype=torch.float64, device=DEV)
    rb = torch.zeros(32, dtype=torch.float64, device=DEV)
    for i in range(0, B, 8192):
        w_, b_ = _wgrad_ref(x[i:i + 8192], dy[i:i + 8192], 3)
        ref += w_
        rb += b_
    # sums of 3.2 M products of unit normals: |dW| ~ 1.8e3; fp32 partial sums of ~3 600 terms, then 1 024 partials
    assert (dW.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    assert (db.double() - rb).abs().max().item() < 2e-5 * max(rb.abs().max().item(), 1.0) * 10
    # shapes outside the kernel's range are reported, not computed wrongly
    assert _ext.conv_wgrad(torch.zeros(2, 24, 7, 7, device=DEV), torch.zeros(2, 32, 7, 7, device=DEV), 3) is None
    assert _ext.conv_wgrad(torch.zeros(2, 16, 9, 9, device=DEV), torch.zeros(2, 16, 9, 9, device=DEV), 3) is None
    assert _ext.conv_wgrad(torch.zeros(2, 64, 8, 8, device=DEV), torch.zeros(2, 64, 8, 8, device=DEV), 3) is None
    assert _ext.conv_wgrad(torch.zeros(2, 16, 5, 1, device=DEV), torch.zeros(2, 16, 5, 1, device=DEV), 3) is None      # (W == 1)


@pytest.mark.gpu
@pytest.mark.parametrize("C,P,B,act", [(32, 49, 300, (1, 0.0)), (32, 64, 7, None), (16, 49, 65536, (1, 0.1)), (48, 64, 33, (1, 0.0)),
                                       (5, 6, 9, None)])
def test_layernorm_channels_bwd_vs_fp64(C, P, B, act):
    from usflows_amd import _ext
    g = torch.Generator().manual_seed(C + P + B)
    x = torch.randn(B, C, P, generator=g).to(DEV)
    dy = torch.randn(B, C, P, generator=g).to(DEV)
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).to(DEV)
    eps = 1e-5
    a = act if act is not None else (_ext.ACT_NONE, 0.0)
    dx, dg, dbt = _ext.layernorm_channels_bwd(x, dy, gamma, eps, a[0], a[1])
    x64 = x.double().requires_grad_(True)
    g64 = gamma.double().requires_grad_(True)
    b64 = torch.zeros(C, dtype=torch.float64, device=DEV, requires_grad=True)
    v = F.leaky_relu(x64, act[1]) if act is not None else x64
    mean = v.mean(dim=1, keepdim=True)
    var = v.var(dim=1, unbiased=False, keepdim=True)
    y = (v - mean) / torch.sqrt(var + eps) * g64.view(1, C, 1) + b64.view(1, C, 1)
    y.backward(dy.double())
    _close(dx, x64.grad, 1e-5, "dx")
    _close(dg, g64.grad, 1e-5, "dgamma")
    _close(dbt, b64.grad, 1e-5, "dbeta")


@pytest.mark.gpu
def test_gated_residual_bwd_and_masked_product():
    from usflows_amd import _ext
    g = torch.Generator().manual_seed(4)
    B, C, P = 37, 32, 49
    x = torch.randn(B, C, P, generator=g).to(DEV)
    vg = torch.randn(B, 2 * C, P, generator=g).to(DEV)
    dy = torch.randn(B, C, P, generator=g).to(DEV)
    dvg = _ext.gated_residual_bwd(dy, vg)
    v64 = vg.double().requires_grad_(True)
    val, gate = v64.chunk(2, dim=1)
    (x.double() + val * torch.sigmoid(gate)).backward(dy.double())
    _close(dvg, v64.grad, 1e-6, "dvg")
    om = (torch.rand(C * P, generator=g) < 0.5).float().to(DEV)
    out = _ext.masked_residual(None, dy, om, -1.0)
    assert torch.equal(out, -(om.view(1, C, P) * dy))


@pytest.mark.gpu
@pytest.mark.parametrize("cout,cin,ks", [(32, 16, 3), (16, 32, 3), (48, 32, 3), (64, 32, 1), (5, 3, 3), (7, 9, 1)])
def test_weight_planes_kernel_gives_the_bits_of_the_torch_split(cout, cin, ks):
    from usflows_amd import _ext
    w = torch.randn(cout, cin, ks, ks, generator=torch.Generator().manual_seed(cout + cin)) * 3
    w[0, 0, 0, 0] = 0.0
    for transposed in (False, True):
        ref = _ext.conv2d_weight_planes(w, transposed=transposed)                       # CPU tensor: the torch formulation
        got = _ext.conv2d_weight_planes(w.to(DEV), transposed=transposed)                # device tensor: one launch
        assert got.shape == ref.shape and got.dtype == torch.bfloat16
        assert torch.equal(got.cpu().view(torch.int16), ref.view(torch.int16))
        assert torch.equal(ref.float().sum(0)[: (cin if transposed else cout)].double().abs().sum() > 0, torch.tensor(True))
    # both orientations from ONE launch (transposed = 2): the same bits
    pf, pt = _ext.conv2d_weight_planes_pair(w.to(DEV))
    for got, transposed in ((pf, False), (pt, True)):
        ref = _ext.conv2d_weight_planes(w, transposed=transposed)
        assert got.shape == ref.shape and torch.equal(got.cpu().view(torch.int16), ref.view(torch.int16))


@pytest.mark.gpu
def test_weight_planes_batch_gives_the_bits_of_the_single_launches():
    from usflows_amd import _ext
    g = torch.Generator().manual_seed(11)
    ws = [torch.randn(co, ci, k, k, generator=g).to(DEV) for co, ci, k in [(32, 16, 3), (32, 32, 3), (16, 32, 3), (64, 32, 1), (5, 3, 3),
                                                                          (48, 7, 3)]]
    batch = _ext.WeightPlanesBatch(ws)
    for _ in range(2):                                          # (the second run reuses the device table)
        out = batch.run()
        assert len(out) == len(ws)
        for w, (pf, pt) in zip(ws, out):
            rf, rt = _ext.conv2d_weight_planes_pair(w)
            assert pf.shape == rf.shape and pt.shape == rt.shape
            assert torch.equal(pf.view(torch.int16), rf.view(torch.int16)) and torch.equal(pt.view(torch.int16), rt.view(torch.int16))
        ws[0].mul_(1.5)                                         # the table names addresses: new values, same table


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,HW,B", [(32, 32, (7, 7), 300), (32, 16, (7, 7), 65), (16, 32, (8, 8), 33), (32, 32, (8, 8), 4099)])
def test_gated_data_gradient_convolution_is_conv_then_gate_then_mask(cin, cout, HW, B):
    """usf_conv2d_same_gate_f32 == usf_conv2d_same_f32, then usf_act_grad_f32 on the input tensor, then the mask product:
    same bits (the factors join the output stream of the same accumulators)"""
    from usflows_amd import _ext
    g = torch.Generator().manual_seed(cin + cout + B)
    H, W = HW
    dy = torch.randn(B, cin, H, W, generator=g).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).to(DEV)
    h = torch.randn(B, cout, H, W, generator=g).to(DEV)
    h[0, 0, 0, 0] = 0.0
    mask = (torch.rand(cout * H * W, generator=g) < 0.5).float().to(DEV)
    planes = _ext.conv2d_weight_planes(w)
    for slope, mul in ((0.0, None), (0.1, mask), (1.0, mask)):
        got = _ext.conv2d_same_gate(dy, planes, cout, 3, h, slope, mul)
        assert got is not None, "shape not served by the fused form"
        ref = _ext.conv2d_same(dy, planes, cout, 3)
        n = ref.numel() // B
        _ext.act_grad(ref, h, M=B, H=n, ldd=n, ldh=n, act=_ext.ACT_LEAKY_RELU, slope=slope)
        if mul is not None:
            ref = _ext.masked_residual(None, ref, mul, 1.0)
        assert torch.equal(got, ref), (slope, mul is not None, (got - ref).abs().max().item())


@pytest.mark.gpu
def test_pointwise_data_gradient_with_the_gate_in_its_pass():
    """usf_pointwise_conv_f32 with out_act = USF_ACT_GATE == the plain pass followed by usf_act_grad_f32 on the gate tensor"""
    from usflows_amd import _ext
    g = torch.Generator().manual_seed(11)
    B, cin, cout, H, W = 301, 64, 32, 7, 7
    dy = torch.randn(B, cin, H, W, generator=g).to(DEV)
    w = (torch.randn(cout, cin, generator=g) * 0.1).to(DEV)
    h = torch.randn(B, cout, H, W, generator=g).to(DEV)
    h[0, 0, 0, 0] = 0.0
    for slope in (0.0, 0.1):
        got = _ext.pointwise_conv(dy, w, out_act=_ext.ACT_GATE, out_slope=slope, gate_x=h)
        ref = _ext.pointwise_conv(dy, w)
        n = ref.numel() // B
        _ext.act_grad(ref, h, M=B, H=n, ldd=n, ldh=n, act=_ext.ACT_LEAKY_RELU, slope=slope)
        assert got.shape == ref.shape and torch.equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("C,P,B,ln,in_act,post_act,bias", [
    (32, 49, 32, True, (1, 0.0), (1, 0.0), True),        # the live MNIST configuration's layer at its training batch
    (32, 49, 300, True, (1, 0.1), (1, 0.2), True),       # > 64 blocks: two rounds of the parameter sums
    (16, 20, 7, True, (1, 0.0), None, False),            # ragged: 140 pixels, no bias, layer norm without a nonlinearity
    (24, 64, 3, False, (1, 0.0), None, True),            # no layer norm: y = x + val * sigmoid(gate)
    (8, 6, 1, True, None, (1, 0.0), True),
])
def test_gated_tail_forward_and_backward_vs_fp64(C, P, B, ln, in_act, post_act, bias):
    """usf_gated_tail_f32 / usf_gated_tail_bwd_f32 against fp64 autograd of the chain they replace (reference networks.py:108-122,
    40-58: 1 x 1 convolution, gate, skip connection, nonlinearity, LayerNormChannels)"""
    from usflows_amd import _ext
    g = torch.Generator().manual_seed(C + P + B)
    h = torch.randn(B, C, P, generator=g).to(DEV)
    x = torch.randn(B, C, P, generator=g).to(DEV)
    dy = torch.randn(B, C, P, generator=g).to(DEV)
    W = (torch.randn(2 * C, C, generator=g) / C ** 0.5).to(DEV)
    bvec = (0.3 * torch.randn(2 * C, generator=g)).to(DEV) if bias else None
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.2 * torch.randn(C, generator=g)).to(DEV)
    eps = 1e-5
    ia = in_act if in_act is not None else (_ext.ACT_NONE, 0.0)
    pa = post_act if post_act is not None else (_ext.ACT_NONE, 0.0)
    lnp = (gamma, beta, eps) if ln else None
    y = _ext.gated_tail(h, x, W, bvec, ia[0], ia[1], pa[0], pa[1], lnp)
    dx, dh, dW, db, dg, dbt, dvg = _ext.gated_tail_bwd(h, x, dy, W, bvec, ia[0], ia[1], pa[0], pa[1], lnp, want_dvg=True)
    assert _ext.gated_tail_bwd(h, x, dy, W, bvec, ia[0], ia[1], pa[0], pa[1], lnp)[6] is None
    h64, x64 = h.double().requires_grad_(True), x.double().requires_grad_(True)
    W64 = W.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.leaky_relu(h64, in_act[1]) if in_act is not None else h64
    vg = torch.einsum("oc,bcp->bop", W64, a)
    if bias:
        vg = vg + bvec.double().view(1, -1, 1)
    vg.retain_grad()
    val, gate = vg.chunk(2, dim=1)
    r = x64 + val * torch.sigmoid(gate)
    if ln:
        v = F.leaky_relu(r, post_act[1]) if post_act is not None else r
        mean = v.mean(dim=1, keepdim=True)
        var = v.var(dim=1, unbiased=False, keepdim=True)
        r = (v - mean) / torch.sqrt(var + eps) * g64.view(1, C, 1) + b64.view(1, C, 1)
    r.backward(dy.double())
    _close(y, r.detach(), 1e-5, "y")
    _close(dx, x64.grad, 1e-5, "dx")
    _close(dh, h64.grad, 1e-5, "dh")
    _close(dvg, vg.grad, 1e-5, "dvg")
    if ln:
        _close(dg, g64.grad, 1e-5, "dgamma")
        _close(dbt, b64.grad, 1e-5, "dbeta")
    else:
        assert dg is None and dbt is None
    _close(dW, W64.grad, 1e-5, "dW")
    _close(db, vg.grad.sum(dim=(0, 2)), 1e-5, "dbias")


# ---- the autograd functions against fp64 torch autograd of the same module ------------------------------------------------
def _grads_of(module_fn, params, x, dy):
    for p in params:
        p.grad = None
    xx = x.clone().requires_grad_(True)
    y = module_fn(xx)
    y.backward(dy)
    return y.detach(), xx.grad, [p.grad.clone() for p in params]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(c_in=16, c_hidden=32, num_layers=1, gating=True, normalize_layers=True, HW=(7, 7)),
                                 dict(c_in=48, c_hidden=32, num_layers=3, gating=True, normalize_layers=True, HW=(8, 8)),
                                 dict(c_in=16, c_hidden=32, num_layers=2, gating=False, normalize_layers=True, HW=(7, 7)),
                                 dict(c_in=16, c_hidden=32, num_layers=1, gating=False, normalize_layers=False, HW=(7, 7)),
                                 dict(c_in=16, c_hidden=32, num_layers=1, gating=True, normalize_layers=False, HW=(5, 4), leaky=0.1)])
def test_convnet2d_trains_on_device_like_fp64_autograd(cfg, monkeypatch):
    import copy
    from usflows_amd import _ext
    from usflows_amd.networks import ConvNet2D
    torch.manual_seed(7)
    H, W = cfg["HW"]
    act = torch.nn.LeakyReLU(cfg["leaky"]) if "leaky" in cfg else torch.nn.ReLU()
    net = ConvNet2D(c_in=cfg["c_in"], c_hidden=cfg["c_hidden"], num_layers=cfg["num_layers"], nonlinearity=act, padding="same",
                    kernel_size=3, normalize_layers=cfg["normalize_layers"], gating=cfg["gating"])
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    ref = copy.deepcopy(net).double()
    net = net.to(DEV)
    B = 19
    x = torch.randn(B, cfg["c_in"], H, W)
    dy = torch.randn(B, cfg["c_in"], H, W)
    mask = (torch.rand(cfg["c_in"] * H * W) < 0.5).float()
    wg = []
    real = _ext.conv_wgrad
    monkeypatch.setattr(_ext, "conv_wgrad", lambda *a_, **k_: (wg.append(1), real(*a_, **k_))[1])
    real_gt = _ext.gated_tail_bwd       # (a GatedConv's 1 x 1 convolution: its weight gradient comes from the fused tail's backward)
    monkeypatch.setattr(_ext, "gated_tail_bwd", lambda *a_, **k_: (wg.append(1), real_gt(*a_, **k_))[1])
    assert net.train_on_device(x.to(DEV).requires_grad_(True))
    y, dx, gp = _grads_of(lambda t: net(t, in_mul=mask.to(DEV)), list(net.parameters()), x.to(DEV), dy.to(DEV))
    n_convs = sum(1 for m in net.modules() if isinstance(m, torch.nn.Conv2d))
    assert len(wg) == n_convs, "a convolution's weight gradient did not come from usf_conv_wgrad_f32 / usf_gated_tail_bwd_f32"
    y64, dx64, gp64 = _grads_of(lambda t: ref(t * mask.double().view(1, cfg["c_in"], H, W)), list(ref.parameters()), x.double(), dy.double())
    _close(y, y64, 2e-5, "y")
    _close(dx, dx64, 2e-5, "dx")
    for (k, _), a_, b_ in zip(net.named_parameters(), gp, gp64):
        _close(a_, b_, 2e-5, k)


@pytest.mark.gpu
@pytest.mark.parametrize("C,HW", [(16, (7, 7)), (48, (8, 8))])
def test_channel_affine_function_vs_fp64(C, HW):
    from usflows_amd.image_training import ChannelAffine
    g = torch.Generator().manual_seed(C)
    B = 21
    x = torch.randn(B, C, *HW, generator=g)
    dy = torch.randn(B, C, *HW, generator=g)
    Wm = torch.randn(C, C, generator=g) / C ** 0.5
    b = torch.randn(C, generator=g)
    for pre_sub in (False, True):
        xd, Wd, bd = x.to(DEV).requires_grad_(True), Wm.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        y = ChannelAffine.apply(xd, Wd, bd, pre_sub)
        y.backward(dy.to(DEV))
        x6, W6, b6 = x.double().requires_grad_(True), Wm.double().requires_grad_(True), b.double().requires_grad_(True)
        y6 = F.conv2d(x6 - b6.view(1, C, 1, 1), W6.view(C, C, 1, 1)) if pre_sub else F.conv2d(x6, W6.view(C, C, 1, 1), b6)
        y6.backward(dy.double())
        _close(y, y6, 1e-5, "y")
        _close(xd.grad, x6.grad, 1e-5, "dx")
        _close(Wd.grad, W6.grad, 1e-5, "dW")
        _close(bd.grad, b6.grad, 1e-5, "db")


# ---- whole flows against the real reference -----------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", image_grad_case_names())
def test_image_flow_device_gradients_match_the_real_reference(name, monkeypatch):
    from usflows_amd import _ext
    flow, a = load_image_case(name, device=DEV)
    loss_ref, g_ref = load_image_grads(name)
    wg = []
    real = _ext.conv_wgrad
    monkeypatch.setattr(_ext, "conv_wgrad", lambda *a_, **k_: (wg.append(1), real(*a_, **k_))[1])
    real_gt = _ext.gated_tail_bwd       # (a GatedConv's 1 x 1 convolution: its weight gradient comes from the fused tail's backward)
    monkeypatch.setattr(_ext, "gated_tail_bwd", lambda *a_, **k_: (wg.append(1), real_gt(*a_, **k_))[1])
    lp = flow.log_prob(a["x"].to(DEV))
    _close(lp.detach(), a["log_prob64"], 1e-5, "log_prob under autograd")
    loss = -lp.mean()
    loss.backward()
    assert abs(float(loss.detach()) - loss_ref) < 1e-5 * abs(loss_ref)
    n_convs = sum(1 for m in flow.modules() if isinstance(m, torch.nn.Conv2d))
    n_aff = sum(1 for l in flow.layers if type(l).__name__ in ("BlockAffineTransform", "InverseTransform"))
    # (runs of consecutive affine layers are composed: K + 1 channel-affine passes for the 2 K + 1 affine layers of a conjugated flow)
    conj = any(type(l).__name__ == "InverseTransform" for l in flow.layers)
    n_aff_runs = (n_aff - 1) // 2 + 1 if conj else n_aff
    assert len(wg) == n_convs + n_aff_runs, f"{len(wg)} weight-gradient launches for {n_convs} convolutions + {n_aff} affine layers"
    named = dict(flow.named_parameters())
    grads_close(named, g_ref)


@pytest.mark.gpu
def test_image_flow_fit_on_device_matches_reference_run(monkeypatch):
    """Flow.fit of the MNIST experiment model on the device reproduces the reference's own run (2 epochs of SGD on 96
    rows): same shuffling, same losses, same parameters after 6 steps -- with the convolutions' gradients on the HIP
    kernels, eager and as a replayed hipGraph"""
    from usflows_amd import _ext
    name = "image_mnistcfg_c16_7x7_k2_gated_ln_hh1_conj"
    data, losses_ref, sd_ref = load_image_fit(name)
    for graph in ("0", "1"):
        monkeypatch.setenv("USFLOWS_AMD_TRAIN_GRAPH", graph)
        flow, _ = load_image_case(name, device=DEV)
        wg = []
        real = _ext.conv_wgrad
        monkeypatch.setattr(_ext, "conv_wgrad", lambda *a_, **k_: (wg.append(1), real(*a_, **k_))[1])
        ds = torch.utils.data.TensorDataset(data, torch.zeros(data.shape[0]))
        np.random.seed(5)
        losses = flow.fit(ds, optim=torch.optim.SGD, optim_params=dict(lr=1e-3), batch_size=32, shuffle=True,
                          device=torch.device(DEV), epochs=2)
        monkeypatch.setattr(_ext, "conv_wgrad", real)
        assert len(wg) > 0, "Flow.fit did not take the device training path"
        for l, r in zip(losses, losses_ref):
            assert abs(float(l) - r) < 2e-4 * abs(r), (graph, losses, losses_ref)
        sd = flow.state_dict()
        for k, v in sd_ref.items():
            s = max(v.abs().max().item(), 1e-3)
            assert (sd[k].cpu().double() - v.double()).abs().max().item() < 2e-3 * s, (graph, k)


@pytest.mark.gpu
@pytest.mark.parametrize("C,nvs,n", [(16, 1, 6), (48, 0, 5), (64, 2, 3), (7, 1, 2), (1, 0, 1), (32, 3, 2)])
def test_affine_prep_kernels_match_the_torch_formulation(C, nvs, n, monkeypatch):
    """usf_affine_prep_f32 / usf_affine_prep_bwd_f32 (one launch each for all blocks of a flow) against the batched torch
    formulation of the same maps in fp64 on the CPU -- matrix, inverse, bias, log|det| and, from random cotangents on all
    four, the gradients of L_raw, U_raw, bias_vector and the Householder vectors"""
    import copy
    from usflows_amd import image_training as it, transforms as T
    g = torch.Generator().manual_seed(100 * C + nvs)
    parts_list = []
    for _ in range(n):
        lu = T.LUTransform(C)
        with torch.no_grad():
            lu.L_raw.copy_(torch.eye(C) + 0.1 * torch.randn(C, C, generator=g).tril(-1))
            sign = torch.where(torch.rand(C, generator=g) < 0.3, -1.0, 1.0)
            lu.U_raw.copy_(torch.diag(sign * (0.75 + 0.5 * torch.rand(C, generator=g))) + 0.1 * torch.randn(C, C, generator=g).triu(1))
            lu.bias_vector.copy_(torch.randn(C, generator=g))
        parts = [lu]
        if nvs:
            hh = T.HouseholderTransform(C, nvs)
            with torch.no_grad():
                hh.vk_householder.copy_(torch.randn(nvs, C, generator=g))
            parts.append(hh)
        parts_list.append(parts)
    sig = (("lu", C),) + ((("hh", C, nvs),) if nvs else ())
    cot = [torch.randn(n, C, C, generator=g), torch.randn(n, C, C, generator=g), torch.randn(n, C, generator=g),
           torch.randn(n, generator=g), torch.randn(n, C, generator=g)]

    def run(pl, device, dtype):
        keys = [object() for _ in pl]
        out = it._prep_group(keys, pl, sig, device)
        loss = 0
        res = []
        for i, k in enumerate(keys):
            o = out[id(k)]
            M, Minv, b, ladj = o[0], o[1], o[2], o[3]
            c = o[4] if len(o) > 4 else -(Minv @ b)              # the kernel also hands out c = -M^-1 b
            loss = loss + (M * cot[0][i].to(device, dtype)).sum() + (Minv * cot[1][i].to(device, dtype)).sum() \
                + (b * cot[2][i].to(device, dtype)).sum() + ladj * cot[3][i].to(device, dtype) \
                + (c * cot[4][i].to(device, dtype)).sum()
            if len(o) <= 5:
                loss = loss + ladj * cot[3][i].to(device, dtype)   # (the device path adds this through the stacked log|det|)
            res.append(tuple(t.detach().cpu().double() for t in (M, Minv, b, ladj, c)))
        o0 = out[id(keys[0])]
        if len(o0) > 5:
            stack = it.prep_stack(o0[5][0])
            assert stack.shape == (len(keys),) and o0[5][1] == 0
            loss = loss + (stack * cot[3].to(device, dtype)).sum()
        loss.backward()
        return res

    ref_pl = [[copy.deepcopy(m).double() for m in parts] for parts in parts_list]
    monkeypatch.setattr(it, "affine_prep_kernels", False)
    ref = run(ref_pl, torch.device("cpu"), torch.float64)
    monkeypatch.setattr(it, "affine_prep_kernels", True)
    dev_pl = [[copy.deepcopy(m).to(DEV) for m in parts] for parts in parts_list]
    assert it._prep_group_device([object() for _ in dev_pl], dev_pl, sig, torch.device(DEV)) is not None
    for parts in dev_pl:
        for m in parts:
            for p in m.parameters():
                p.grad = None
    got = run(dev_pl, torch.device(DEV), torch.float32)

    def close(a, b, what, tol=2e-5):
        scale = max(b.abs().max().item(), 1e-6)
        assert (a - b).abs().max().item() <= tol * scale * max(C, 8) ** 0.5, (what, (a - b).abs().max().item(), scale)

    for i in range(n):
        for q, what in enumerate(("M", "Minv", "b", "ladj", "c")):
            close(got[i][q], ref[i][q], (what, i))
        lu_d, lu_r = dev_pl[i][0], ref_pl[i][0]
        close(lu_d.L_raw.grad.cpu().double(), lu_r.L_raw.grad, ("dL_raw", i), 1e-4)
        close(lu_d.U_raw.grad.cpu().double(), lu_r.U_raw.grad, ("dU_raw", i), 1e-4)
        close(lu_d.bias_vector.grad.cpu().double(), lu_r.bias_vector.grad, ("dbias", i), 1e-4)
        if nvs:
            close(dev_pl[i][1].vk_householder.grad.cpu().double(), ref_pl[i][1].vk_householder.grad, ("dvk", i), 1e-4)


@pytest.mark.gpu
def test_deferred_partial_sums_of_a_backward_pass_are_one_launch_and_the_same_bits(monkeypatch):
    """small batches: the last stage of every convolution weight gradient of a backward pass (the sum over per-wave partial
    slots) is queued and leaves as ONE usf_partial_sum_jobs_f32 launch when the pass ends -- same additions in the same order
    as usf_conv_wgrad_f32's own sum: same bits; a parameter that already holds a gradient keeps the undeferred call"""
    from usflows_amd import _ext
    flow, a = load_image_case("image_mnistcfg_c16_7x7_k2_gated_ln_hh1_conj", device=DEV)
    x = a["x"].to(DEV)
    launches = []
    real = _ext._launch
    monkeypatch.setattr(_ext, "_launch", lambda name, *a_, **k_: (launches.append(name), real(name, *a_, **k_))[1])

    def grads(env, scope=True, twice=False):
        from usflows_amd.config import config
        monkeypatch.setattr(config, "psum_jobs", env == "1")
        for p in flow.parameters():
            p.grad = None
        del launches[:]
        loss = -flow.log_prob(x).mean()
        if twice:                                   # the same weights feed two nodes of ONE backward graph
            loss = loss - flow.log_prob(x.flip(0)).mean()
        import contextlib
        with (_ext.deferred_sums_scope() if scope else contextlib.nullcontext()):      # (Flow.fit opens this scope around its own steps)
            loss.backward()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in flow.named_parameters() if p.grad is not None}, launches.count("usf_partial_sum_jobs_f32")

    g_off, n_off = grads("0")
    w0 = _ext.n_wgrad_jobs_flushed[0]
    g_on, n_on = grads("1")
    assert n_off == 0 and 1 <= n_on <= 2, (n_off, n_on)
    # ... and the 3 x 3 weight-gradient kernels themselves left as one usf_conv_wgrad_jobs_f32 launch per tile shape
    n_conv3 = sum(1 for m in flow.modules() if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3))
    assert _ext.n_wgrad_jobs_flushed[0] - w0 == n_conv3 and 1 <= launches.count("usf_conv_wgrad_jobs_f32") <= 4
    assert set(g_on) == set(g_off)
    for k in g_off:
        assert torch.equal(g_on[k], g_off[k]), k
    # outside a scope (any loss.backward() that is not Flow.fit's own: hooks / DistributedDataParallel may read gradients mid-pass)
    g_out, n_out = grads("1", scope=False)
    assert n_out == 0 and all(torch.equal(g_out[k], g_off[k]) for k in g_off)
    # two log_prob calls in one loss -- autograd adds the two gradients of a weight as soon as the second arrives: the
    # queue is flushed in front of that and the second producer is not deferred; same bits as without any deferral
    g2_off, _ = grads("0", twice=True)
    g2_on, _n2 = grads("1", twice=True)
    assert set(g2_on) == set(g2_off)
    for k in g2_off:
        # (a parameter that receives more than two contributions in this graph -- an affine block of a conjugated flow, used as M
        # and M^-1 by each of the two passes -- is summed by autograd in an order that is not fixed: values, not bits)
        assert torch.allclose(g2_on[k], g2_off[k], rtol=2e-5, atol=2e-6 * float(g2_off[k].abs().max())), k
        if "conditioner" in k:
            assert torch.equal(g2_on[k], g2_off[k]), k
    # gradients already in place: autograd adds the new ones at once -> no PARAMETER's gradient may be deferred (what may still be
    # queued: the C x C gradients of the composed affine runs, whose consumer -- image_training.RunsOut / AffinePrep -- issues the
    # queue before it reads them: at most one per affine layer)
    g_on, n_on = grads("1")
    del launches[:]
    flushed0 = _ext.n_jobs_flushed[0]
    with _ext.deferred_sums_scope():
        (-flow.log_prob(x).mean()).backward()
    torch.cuda.synchronize()
    n_aff = sum(1 for l in flow.layers if type(l).__name__ in ("BlockAffineTransform", "InverseTransform"))
    assert launches.count("usf_partial_sum_jobs_f32") <= 2 and _ext.n_jobs_flushed[0] - flushed0 <= n_aff
    named = dict(flow.named_parameters())
    for k in g_off:
        assert torch.allclose(named[k].grad, 2 * g_off[k], rtol=1e-5, atol=1e-6 * float(g_off[k].abs().max())), k
