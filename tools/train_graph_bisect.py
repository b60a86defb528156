"""which part of a composite training step survives hipGraph capture (each variant in a child process: a failing
hipStreamEndCapture takes the process down).  python tools/train_graph_bisect.py            (tuning aid)"""
import os
import subprocess
import sys

VARIANTS = ["direct_sophia", "directdel_sophia", "fit_sophia", "fit_sgd"]

if len(sys.argv) > 1:
    import warnings
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from usflows_amd.flows import USFlow, _unvalidated
    from usflows_amd.networks import ConvNet2D
    from usflows_amd.sophia import SophiaG
    from usflows_amd import transforms as T
    warnings.simplefilter("ignore")
    v = sys.argv[1]
    if "_d" in v:
        os.environ["USF_TG_DBG"] = v.split("_d")[1]
    dev = "cuda:0"
    dims = [16, 7, 7]
    torch.manual_seed(0)
    base = torch.distributions.Laplace(torch.zeros(dims).to(dev), torch.ones(dims).to(dev))
    flow = USFlow(base, dims, 2, ConvNet2D, dict(c_in=16, c_hidden=32, num_layers=1, padding="same", kernel_size=3,
                                                normalize_layers=True, gating=True, nonlinearity=torch.nn.ReLU()),
                  householder=1, affine_conjugation=True).to(dev)
    x = torch.rand(32, *dims, device=dev)
    params = list(flow.parameters())
    opt = SophiaG(params, lr=1e-5) if "sophia" in v else torch.optim.SGD(params, lr=1e-5)

    if v == "lu_only_bwd":
        lu = T.LUTransform(16).to(dev)
        params = list(lu.parameters())

        def loss_fn():
            return lu.backward(x.permute(0, 2, 3, 1).reshape(-1, 16)).sum() + lu.inverse_matrix().sum()
    elif v == "hh_only_bwd":
        hh = T.HouseholderTransform(16, 1).to(dev)
        params = list(hh.parameters())

        def loss_fn():
            return hh.matrix().sum() + hh.inverse_matrix().sum()
    elif v == "conv_only_bwd":
        net = flow.layers[1].conditioner
        params = list(net.parameters())

        def loss_fn():
            return net(x).sum()
    else:
        def loss_fn():
            with _unvalidated(flow.base_distribution):
                return -flow.log_prob(x).mean() - flow.log_prior()

    def step():
        for p in params:
            if p.grad is not None:
                p.grad.zero_()
        loss = loss_fn()
        if "bwd" in v:
            loss.backward()
        if v.endswith("sgd") or v.endswith("sophia"):
            opt.step()
        return loss.detach()

    if v.startswith("direct"):
        if "to_" in v:
            flow = flow.to(torch.device(dev))
            opt = SophiaG(flow.parameters(), lr=1e-6)
        xh = torch.rand(32, *dims)
        for i in range(6):
            xs = xh.to(dev) if "host" in v else x
            r = flow._train_graph_step(opt, xs, None)
            if r is None:
                opt.zero_grad()
                loss = -flow.log_prob(xs).mean() - flow.log_prior()
                loss.backward()
                opt.step()
                if "del" in v:
                    del loss
            if "feas" in v:
                assert flow.is_feasible()
                flow.transform.clear_cache()
        print(f"{v}: ran, replays {flow._train_graph_state['replays']}", flush=True)
        os._exit(0)
    if v.startswith("fit"):
        import copy
        f = copy.deepcopy(flow) if "clone" in v else flow
        N = 8 * 32
        data = torch.utils.data.TensorDataset(torch.rand(N, *dims), torch.zeros(N))
        losses = f.fit(data, optim=SophiaG if "sophia" in v else torch.optim.SGD, optim_params=dict(lr=1e-6), batch_size=32,
                       device=torch.device(dev), epochs=1)
        print(f"{v}: fit ran, replays {f._train_graph_state['replays']}, loss {losses}", flush=True)
        os._exit(0)
    if v.startswith("deepcopy"):
        import copy
        flow = copy.deepcopy(flow)
        params = list(flow.parameters())
        opt = SophiaG(params, lr=1e-5)
    for _ in range(3):
        if v.startswith("setnone"):
            opt.zero_grad()
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    g.replay()
    torch.cuda.synchronize()
    print(f"{v}: captured and replayed, loss {float(out):.4f}", flush=True)
    os._exit(0)

for v in VARIANTS:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), v], capture_output=True, text=True, timeout=240)
    tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and not l.startswith(("Search", "HIP kernel", "For debugging", "Compile", "/opt/amdgpu"))]
    print(f"[{v}] rc={r.returncode}: {tail[-1][:200] if tail else ''}", flush=True)
