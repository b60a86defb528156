"""would a hipGraph of the image flow's layer loop pay at small batches? (tuning aid)"""
import sys, os, time, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd.flows import USFlow
from usflows_amd.networks import ConvNet2D
warnings.simplefilter("ignore")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = "cuda:0"; dims = [16, 7, 7]
base = torch.distributions.Laplace(torch.zeros(dims).to(dev), torch.ones(dims).to(dev))
flow = USFlow(base, dims, 2, ConvNet2D, dict(c_in=16, c_hidden=32, num_layers=1, padding="same", kernel_size=3, normalize_layers=True,
                                            gating=True, nonlinearity=torch.nn.ReLU()), householder=1, affine_conjugation=True).to(dev)
x = torch.rand(B, *dims, device=dev)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    eager = t(lambda: flow.log_prob(x))
    ref = flow.log_prob(x).clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): flow.log_prob(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = flow.log_prob(x)
    graph = t(lambda: g.replay())
    g.replay(); torch.cuda.synchronize()
    print(f"B={B}: eager {eager:.3f} ms, graph replay {graph:.3f} ms, equal: {torch.equal(out, ref)}")
