"""log_prob of a flat flow with the reference's default vector ConvNet conditioner (GatedMLP + LayerNormVector): engine
(chain of usf_linear_f32 + usf_gated_norm_rows_f32 ops) vs the torch composite loop of the same modules, ms per call.
    python tools/gated_vec_probe.py [rows] [blocks]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd.synth import ModelSpec, build_usflow, synth_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spec = ModelSpec(784, K, [256, 256], householder=0, conditioner="ConvNet", extra={"gating": True, "normalize_layers": True})
flow = build_usflow(spec, synth_state_dict(spec, seed=3), device="cuda:0")
x = torch.rand(B, 784, device="cuda:0")


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    a = flow.log_prob(x)
    b = flow._layer_loop_log_prob(x)
    print("max rel diff engine vs composite:", ((a - b).abs() / b.abs()).max().item())
    print(f"rows {B} blocks {K}: engine {timed(lambda: flow.log_prob(x)):.2f} ms, "
          f"torch composite {timed(lambda: flow._layer_loop_log_prob(x)):.2f} ms")
