"""debug: first wgrad_blocked call of the planes backward with its real operands vs fp64 of the decoded operands"""
import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_case
from model_util import build_flow
from oracle import usflows_oracle as orc
from usflows_amd.synth import synth_state_dict
from usflows_amd import _ext

def decode(buf, M, nkb):
    npan = -(-M // 16)
    v = buf[: npan * nkb * 3072].view(torch.bfloat16).view(npan, nkb, 3, 4, 16, 8).float().sum(2)      # [npan, nkb, g, j, u]
    out = torch.zeros(npan, 16, nkb, 32, device=buf.device)
    for g in range(4):
        for u in range(8):
            out[:, :, :, 16 * (u >> 2) + 4 * g + (u & 3)] = v[:, :, g, :, u].permute(0, 2, 1)
    return out.reshape(npan * 16, nkb * 32)[:M]

B, slope = int(sys.argv[1]), float(sys.argv[2])
spec, _sd, _a = load_case("synth_d784_k32_cfg2")
spec = copy.copy(spec); spec.coupling_blocks, spec.negative_slope = 4, slope
sd = synth_state_dict(spec, seed=5, alpha=0.1)
g = torch.Generator().manual_seed(B)
x = torch.rand(B, 784, generator=g)
g_lp = -(0.5 + torch.rand(B, generator=g)) / B
flow = build_flow(spec, sd, device="cuda:0")
eng = flow.engine(); eng.train_planes_min_rows = eng.fused_min_rows = 0
calls = []
aff_ops = []
real = _ext.wgrad_blocked
def spy(Yp, y_nkb, y_kb0, Ap, a_nkb, a_kb0, G, **kw):
    real(Yp, y_nkb, y_kb0, Ap, a_nkb, a_kb0, G, **kw)
    torch.cuda.synchronize()
    M, N, K = kw["M"], kw["N"], kw["K"]
    Y = decode(Yp, M, y_nkb)[:, 32 * y_kb0: 32 * y_kb0 + N].double(); A = decode(Ap, M, a_nkb)[:, 32 * a_kb0: 32 * a_kb0 + K].double()
    ref = kw.get("alpha", 1.0) * (Y.t() @ A)
    got = torch.as_strided(G.reshape(-1), (N, K), (kw["ldg"], 1)).double()
    d = got - ref
    cs = kw.get("colsum")
    cse = None
    if cs is not None:
        cref = kw.get("cs_alpha", 1.0) * Y.sum(0)
        cse = ((cs.reshape(-1)[:N].double() - cref).abs().max() / cref.abs().max()).item()
    calls.append((N, K, (d.abs().max() / ref.abs().max()).item(), (d.norm() / ref.norm()).item(), cse, Y.abs().max().item(), A.abs().max().item()))
    if N == 784 and K == 784:
        aff_ops.append((decode(Yp, M, y_nkb).double().cpu(), decode(Ap, M, a_nkb).double().cpu()))
_ext.wgrad_blocked = spy
lp = flow.log_prob(x.cuda())
(lp * g_lp.cuda()).sum().backward()
torch.cuda.synchronize()
for c in calls:
    print("wgrad N=%d K=%d  max/big %.2e fro %.2e colsum %s  |Y| %.2e |A| %.2e" % (c[0], c[1], c[2], c[3], "%.2e" % c[4] if c[4] is not None else None, c[5], c[6]))
# forward activations vs fp64: latent z
with torch.no_grad():
    for planes in (True, False):
        eng.use_planes = planes; eng.planes_min_rows = 0
        z = flow.backward(x.cuda()).cpu().double()
        sd64 = {k: v.double() for k, v in sd.items()}
        zr = orc.flow_backward(sd64, spec, x.double()) if hasattr(orc, "flow_backward") else None
        if zr is not None:
            print("forward planes" if planes else "forward rows", "z max err / max|z| %.2e  fro %.2e" % (((z - zr).abs().max() / zr.abs().max()).item(), ((z - zr).norm() / zr.norm()).item()))

# fp64 reference of the gradient signal at every affine layer's output / the layer inputs
sd64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
xx = x.double()
outs, ins = [], []
for kind, prefix, flip, seq in reversed(orc.layer_plan(spec)):
    if kind == "scale":
        y = xx / sd64[prefix + "scale"]
    elif kind == "affine":
        ap = orc._AffineParams(sd64, prefix, spec, seq)
        ins.append(xx)
        y = ap.backward(xx)
        y.retain_grad()
        outs.append(y)
    elif kind == "coupling":
        y = orc.coupling_backward(sd64, prefix, spec, orc._mask_for(spec, flip, xx.dtype), xx, None)
    xx = y
lp64 = orc.base_log_prob(spec, xx, sd64)
(lp64 * g_lp.double()).sum().backward()
segp = eng.segp_idx
for i, (Yd, Ad) in enumerate(aff_ops):                  # backward order: last affine first
    gref, aref = outs[len(outs) - 1 - i].grad, ins[len(ins) - 1 - i].detach()
    if i == 0:
        Yn = Yd[:, :784]
    else:
        Yn = torch.zeros_like(gref); ok = segp >= 0; Yn[:, segp[ok]] = Yd[:, ok.nonzero().flatten()]
    An = torch.zeros_like(aref); ok = segp >= 0; An[:, segp[ok]] = Ad[:, ok.nonzero().flatten()]
    if i == len(aff_ops) - 1:
        aref = None          # head: a' = x / s - b
    e = Yn - gref
    print("affine #%d (backward order): g max/big %.2e fro %.2e" % (i, (e.abs().max() / gref.abs().max()).item(), (e.norm() / gref.norm()).item()),
          "" if aref is None else "| a max/big %.2e fro %.2e" % (((An - aref).abs().max() / aref.abs().max()).item(), ((An - aref).norm() / aref.norm()).item()))
    rowerr = e.abs().max(1).values
    colerr = e.abs().max(0).values
    print("    worst rows", rowerr.topk(3).indices.tolist(), ["%.1e" % v for v in rowerr.topk(3).values.tolist()], " worst cols", colerr.topk(5).indices.tolist(), ["%.1e" % v for v in colerr.topk(5).values.tolist()])
