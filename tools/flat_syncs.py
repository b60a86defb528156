"""which calls of the flat (engine) path synchronise the host? (tuning aid: torch.cuda.set_sync_debug_mode)"""
import sys, os, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd.synth import ModelSpec, synth_state_dict, build_usflow
spec = ModelSpec(784, 4, [256, 256], householder=0, affine_conjugation=False, negative_slope=0.01,
                 conditioner="ConditionalDenseNN", base="laplace")
flow = build_usflow(spec, synth_state_dict(spec, seed=100, alpha=0.1), device="cuda:0")
x = torch.rand(int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 784, device="cuda:0")
from usflows_amd.sophia import SophiaG
opt = SophiaG(flow.parameters(), lr=1e-6)
def train():
    opt.zero_grad(set_to_none=True)
    loss = -flow.log_prob(x).mean()
    loss.backward()
    opt.step()
calls = {"log_prob": lambda: flow.log_prob(x), "backward": lambda: flow.backward(x), "_forward": lambda: flow._forward(x),
         "sample": lambda: flow.sample([x.shape[0]], seed=1)}
with torch.no_grad():
    for f in calls.values():
        f(); f()
for _ in range(3):
    train()
torch.cuda.synchronize()
for name, f in list(calls.items()) + [("train step", train)]:
    torch.cuda.set_sync_debug_mode("error")
    try:
        if name == "train step":
            f()
        else:
            with torch.no_grad():
                f()
        print(name, ": no synchronising call")
    except Exception:
        tb = traceback.format_exc().strip().splitlines()
        print(name, ": SYNC at", [l.strip() for l in tb if "usflows_amd" in l or "torch/" in l][-3:])
    finally:
        torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
