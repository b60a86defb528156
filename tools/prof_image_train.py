"""host-side profile of one image-flow training step (python3 tools/prof_image_train.py [batch])"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from usflows_amd.flows import USFlow  # noqa: E402
from usflows_amd.networks import ConvNet2D  # noqa: E402
from usflows_amd.sophia import SophiaG  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = bench.IMAGE_CONFIGS["mnist_image"]
dims = list(cfg["in_dims"])
dev = torch.device("cuda:0")
torch.manual_seed(100)
host = USFlow(torch.distributions.Laplace(torch.zeros(dims), torch.ones(dims)), dims, cfg["blocks"], ConvNet2D, dict(cfg["cond"]),
              householder=cfg["householder"], affine_conjugation=True)
bench._condition_image_flow(host, seed=100)
flow = USFlow(torch.distributions.Laplace(torch.zeros(dims, device=dev), torch.ones(dims, device=dev)), dims, cfg["blocks"],
              ConvNet2D, dict(cfg["cond"]), householder=cfg["householder"], affine_conjugation=True)
flow.load_state_dict(host.state_dict(), strict=True)
flow = flow.to(dev)
x = torch.rand(B, *dims).to(dev)
opt = SophiaG(flow.parameters(), lr=1e-6)


def step():
    opt.zero_grad(set_to_none=True)
    loss = -flow.log_prob(x).mean()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumtime").print_stats(45)
st.sort_stats("tottime").print_stats(25)
