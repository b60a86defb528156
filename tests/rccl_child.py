"""Child process of tests/test_00_rccl_gpu.py: the data-parallel code paths over a REAL "nccl" (= RCCL) process group of ONE
rank on the one GPU of the test box -- communicator initialisation, the fp64 scalar all-reduce of ``mean_log_prob`` on the
compute stream, the flat-arena gradient all-reduce of flat training and the flat-buffer gradient all-reduce of image
training.  Started as a fresh process (nothing here runs in a process that touched the GPU before); writes one JSON object
with its findings to the path given as argv[1]."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(out_path):
    import copy
    import numpy as np
    import torch
    import torch.distributed as dist
    res = {}
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    res["backend"] = dist.get_backend()
    try:
        from golden_util import load_case, load_image_radial_case
        from model_util import build_flow
        from usflows_amd import _ext
        from usflows_amd.parallel import allreduce_gradients, data_parallel_training, mean_log_prob
        res["native_lib"] = _ext.lib_exists()

        # ---- (1) mean_log_prob: flat flow, the collective forced, no host synchronisation ----
        spec, sd, a = load_case("synth_d64_k4_hh1_conj_laplace")
        flow = build_flow(spec, sd, device="cuda:0")
        x = torch.rand(4096, spec.dim, generator=torch.Generator().manual_seed(3)).to(dev)
        x[:a["x"].shape[0]] = a["x"].to(dev)
        n_coll = [0]
        real_ar = dist.all_reduce

        def counting(*args, **kw):
            n_coll[0] += 1
            return real_ar(*args, **kw)

        dist.all_reduce = counting
        with torch.no_grad():
            want = flow.log_prob(x)
        acc = torch.zeros(2, dtype=torch.float64, device=dev)
        mean_log_prob(flow, x, acc=acc, force_collective=True)             # warm (communicator, caches)
        torch.cuda.synchronize()
        c0 = n_coll[0]
        torch.cuda.set_sync_debug_mode("error")
        try:
            mean, lp = mean_log_prob(flow, x, acc=acc, force_collective=True)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        res["mean_collectives_per_call"] = n_coll[0] - c0
        res["mean_equal"] = bool(torch.equal(lp, want)) and abs(float(mean) - float(want.double().mean())) <= 1e-12 * abs(float(mean))
        res["mean_golden_rel"] = float(((lp[:a["x"].shape[0]].double().cpu() - a["log_prob64"]).abs() / a["log_prob64"].abs()).max())
        c0 = n_coll[0]
        mean_log_prob(flow, x, acc=acc)                                     # default: a one-rank group issues nothing
        res["mean_collectives_default_world1"] = n_coll[0] - c0

        # ---- (2) flat training: the gradient arena through the all-reduce ----
        ref_flow = build_flow(spec, sd, device="cuda:0")
        lp_r = ref_flow.log_prob(x[:512])
        (-lp_r.mean()).backward()
        data_parallel_training(flow, force_collective=True)
        c0 = n_coll[0]
        lp_d = flow.log_prob(x[:512])
        (-lp_d.mean()).backward()
        res["flat_train_collectives"] = n_coll[0] - c0
        worst = 0.0
        for (k, p), (_, q) in zip(flow.named_parameters(), ref_flow.named_parameters()):
            if q.grad is None:
                continue
            sc = max(q.grad.abs().max().item(), 1e-30)
            worst = max(worst, (p.grad - q.grad).abs().max().item() / sc)
        res["flat_train_grad_max_rel"] = worst            # (weights B / B: one multiply and one divide per entry)

        # ---- (3) image training (the live MNIST configuration): gradients as views of one buffer, one all-reduce per step ----
        name = "imageradial_mnistlive_c16_7x7_k2_l3_lognormal"
        img, arr, g_ref, _ = load_image_radial_case(name, device="cuda:0")
        xi = arr["x"].to(dev)
        c0 = n_coll[0]
        for step in range(2):
            img.__dict__["_grad_allreduce"] = (None, True)
            if step:
                img._zero_grad_for_step(torch.optim.SGD(img.parameters(), lr=0.0))
            loss = -img.log_prob(xi).mean() - img.log_prior()
            loss.backward()
            allreduce_gradients(img, xi.shape[0])
        res["image_train_collectives"] = n_coll[0] - c0
        st = img.__dict__.get("_dp_grads")
        res["image_grads_bound"] = bool(st is not None and all(p.grad.data_ptr() == v.data_ptr() for p, v in
                                                               zip([p for p in img.parameters() if p.requires_grad], st["views"])))
        named = dict(img.named_parameters())
        worst = 0.0
        for k, g in g_ref.items():
            sc = max(g.abs().max().item(), 1e-30)
            worst = max(worst, (named[k].grad.double().cpu() - g).abs().max().item() / sc)
        res["image_train_grad_max_rel_vs_reference"] = worst
        dist.all_reduce = real_ar
        res["ok"] = True
    except Exception as e:          # noqa: BLE001
        import traceback
        res["ok"] = False
        res["error"] = f"{type(e).__name__}: {e}"
        res["trace"] = traceback.format_exc()[-3000:]
    finally:
        try:
            torch.cuda.synchronize()
            dist.destroy_process_group()
        except Exception:           # noqa: BLE001
            pass
    with open(out_path, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main(sys.argv[1])
