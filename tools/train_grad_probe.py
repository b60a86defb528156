"""Per-parameter deviation of the device training gradients from fp64 autograd through the oracle, planes path and fp32-row
path side by side (a debugging / evidence tool: python tools/train_grad_probe.py B slope [blocks])."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_case            # noqa: E402
from model_util import build_flow            # noqa: E402
from oracle import usflows_oracle as orc     # noqa: E402
from usflows_amd.synth import synth_state_dict   # noqa: E402

B, slope = int(sys.argv[1]), float(sys.argv[2])
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 4
spec, _sd, _a = load_case("synth_d784_k32_cfg2")
spec = copy.copy(spec)
spec.coupling_blocks, spec.negative_slope = blocks, slope
sd = synth_state_dict(spec, seed=5, alpha=0.1)
g = torch.Generator().manual_seed(B)
x = torch.rand(B, 784, generator=g)
g_lp = -(0.5 + torch.rand(B, generator=g)) / B
sd64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
lp_ref = orc.flow_log_prob(sd64, spec, x.double())
(lp_ref * g_lp.double()).sum().backward()
res = {}
for planes in (True, False):
    flow = build_flow(spec, sd, device="cuda:0")
    eng = flow.engine()
    eng.use_train_planes = planes
    eng.train_planes_min_rows = eng.fused_min_rows = 0
    lp = flow.log_prob(x.cuda())
    (lp * g_lp.cuda()).sum().backward()
    torch.cuda.synchronize()
    res[planes] = {n: p.grad.detach().cpu().double() for n, p in flow.named_parameters() if p.grad is not None}
    print("planes" if planes else "fp32 rows", "log_prob max rel", float(((lp.detach().cpu().double() - lp_ref.detach()).abs() / lp_ref.detach().abs()).max()))
print(f"{'parameter':62s} {'big':>9s} | planes: max/big  fro | rows: max/big  fro | planes vs rows max/big")
for n in res[True]:
    ref = sd64[n].grad
    if ref is None:
        continue
    ref = ref.reshape(res[True][n].shape)
    big = ref.abs().max().item()
    if big == 0:
        continue
    d1, d0 = res[True][n] - ref, res[False][n] - ref
    print(f"{n:62s} {big:9.2e} | {d1.abs().max().item() / big:8.1e} {d1.norm().item() / ref.norm().item():8.1e} | "
          f"{d0.abs().max().item() / big:8.1e} {d0.norm().item() / ref.norm().item():8.1e} | {(res[True][n] - res[False][n]).abs().max().item() / big:8.1e}")
