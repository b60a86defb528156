"""Flow.fit of an image-shaped flow with and without the graphed training step: ms per step.  (First version of this probe:
can one training step of an image-shaped flow (composite torch formulation under autograd + SophiaG) be captured as a
hipGraph?  eager vs replayed ms per step, and the loss trajectories of both from the same start (tuning aid))"""
import copy
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd.flows import USFlow  # noqa: E402
from usflows_amd.networks import ConvNet2D  # noqa: E402
from usflows_amd.sophia import SophiaG  # noqa: E402

warnings.simplefilter("ignore")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda:0"
dims = [16, 7, 7]
torch.manual_seed(0)
base = torch.distributions.Laplace(torch.zeros(dims).to(dev), torch.ones(dims).to(dev))
flow = USFlow(base, dims, 2, ConvNet2D, dict(c_in=16, c_hidden=32, num_layers=1, padding="same", kernel_size=3, normalize_layers=True,
                                            gating=True, nonlinearity=torch.nn.ReLU()), householder=1, affine_conjugation=True).to(dev)
import numpy as np  # noqa: E402

N = 64 * B
data = torch.utils.data.TensorDataset(torch.rand(N, *dims), torch.zeros(N))
for graph in ((True,) if os.environ.get('PROBE_GRAPH_ONLY') else (False, True)):
    f = copy.deepcopy(flow)
    f.use_train_graph = graph
    np.random.seed(0)
    f.fit(data, optim=SophiaG, optim_params=dict(lr=1e-5), batch_size=B, device=torch.device(dev), epochs=1)     # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = f.fit(data, optim=SophiaG, optim_params=dict(lr=1e-5), batch_size=B, device=torch.device(dev), epochs=2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 128 * 1e3
    st = f.__dict__.get("_train_graph_state")
    print(f"Flow.fit B={B} graph={graph}: {dt:.2f} ms/step (incl. the eager first steps and the host copy of each batch), "
          f"losses {[round(float(v), 4) for v in losses]}, replays {st['replays'] if st else 0}")
