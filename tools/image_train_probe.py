"""one training step of the reference's MNIST image configuration on the GPU (composite torch formulation under autograd;
tuning aid for SURVEY row N4)"""
import sys, os, time, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd.flows import USFlow
from usflows_amd.networks import ConvNet2D
from usflows_amd.sophia import SophiaG
warnings.simplefilter("ignore")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda:0"; dims = [16, 7, 7]
base = torch.distributions.Laplace(torch.zeros(dims).to(dev), torch.ones(dims).to(dev))
flow = USFlow(base, dims, 2, ConvNet2D, dict(c_in=16, c_hidden=32, num_layers=1, padding="same", kernel_size=3, normalize_layers=True,
                                            gating=True, nonlinearity=torch.nn.ReLU()), householder=1, affine_conjugation=True).to(dev)
opt = SophiaG(flow.parameters(), lr=1e-5)
x = torch.rand(B, *dims, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = -flow.log_prob(x).mean()
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n): step()
torch.cuda.synchronize()
print(f"image flow training step B={B}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
