import sys, os, warnings, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd.flows import USFlow
from usflows_amd.networks import ConvNet2D
dev = "cuda:0"; dims = [16, 7, 7]
base = torch.distributions.Laplace(torch.zeros(dims).to(dev), torch.ones(dims).to(dev))
flow = USFlow(base, dims, 2, ConvNet2D, dict(c_in=16, c_hidden=32, num_layers=1, padding="same", kernel_size=3,
                                            normalize_layers=True, gating=True, nonlinearity=torch.nn.ReLU()),
              householder=1, affine_conjugation=True).to(dev)
x = torch.rand(256, *dims, device=dev)
with torch.no_grad():
    flow.log_prob(x); flow.log_prob(x)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        flow.log_prob(x)
        print("no synchronising call")
    except Exception as e:
        traceback.print_exc()
