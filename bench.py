#!/usr/bin/env python3
"""Headline benchmark: log_prob samples/sec of the BASELINE cfg2 flow (D=784, 32 additive coupling
blocks, LeakyReLU MLP conditioner [256,256], Laplace base) at batch 65536 per GPU, fp32.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full Flow.log_prob pass over this rank's resident batch (inputs already in HBM)
+ the scalar mean-log_prob all-reduce.  Prints ONE JSON line (rank 0) with the whole-job
throughput, the roofline of the dominant kernel (HIP-event timed inside the timed region) and a
CPU baseline (the oracle, bounded sample, rank 0 at N=1 only)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0      # same guide: dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="rows per GPU")
    ap.add_argument("--dim", type=int, default=784)
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--hidden", type=int, nargs="+", default=[256, 256])
    ap.add_argument("--householder", type=int, default=0, help="Householder vectors per affine block (USFlow ctor default 1)")
    ap.add_argument("--conj", action="store_true", help="affine_conjugation=True (what the reference's live configs use)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="run couplings as 3 linear ops instead of the fused kernel")
    ap.add_argument("--fused-min-rows", type=int, default=None, help="batch size from which couplings take the fused kernel")
    ap.add_argument("--gemm", choices=["bf16x3", "f32"], default=None,
                    help="affine GEMM arithmetic: bf16x3 = 3-way split on the bf16 MFMA (default), f32 = exact-f32 MFMA")
    ap.add_argument("--mode", choices=["log_prob", "sample", "train"], default="log_prob",
                    help="sample: time Flow.sample (Philox head + forward pass), BASELINE cfg5 per GPU; train: one "
                         "optimiser step of Flow.fit's loss (-log_prob.mean(): device forward + backward + Adam), 1 GPU")
    ap.add_argument("--cpu-rows", type=int, default=4096)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    from usflows_amd.synth import ModelSpec, synth_state_dict, build_usflow
    from usflows_amd.parallel import mean_log_prob

    spec = ModelSpec(args.dim, args.blocks, list(args.hidden), householder=args.householder, affine_conjugation=args.conj,
                        negative_slope=0.01, conditioner="ConditionalDenseNN", base="laplace")
    sd = synth_state_dict(spec, seed=100, alpha=0.1)         # same parameters on every rank
    flow = build_usflow(spec, sd, device=str(dev))
    eng = flow.engine()
    eng.use_fused_coupling = not args.unfused
    if args.fused_min_rows is not None:
        eng.fused_min_rows = args.fused_min_rows
    if args.gemm:
        eng.gemm_mode = args.gemm
    B, D = args.batch, args.dim
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(B, D, generator=g).to(dev)               # this rank's shard, resident in HBM
    acc = torch.zeros(2, dtype=torch.float64, device=dev)

    opt = torch.optim.Adam(flow.parameters(), lr=1e-6) if args.mode == "train" else None
    if args.mode == "train" and distributed:
        from usflows_amd.parallel import data_parallel_training
        data_parallel_training(flow)                       # one all-reduce of the flat gradient arena per step

    def step():
        if args.mode == "train":
            opt.zero_grad(set_to_none=True)
            lp_ = flow.log_prob(x)
            loss = -lp_.mean()
            loss.backward()
            opt.step()
            return -loss.detach().double(), lp_.detach()
        if args.mode == "sample":
            with torch.no_grad():
                xs = flow.sample([B], seed=1234, row_offset=rank * B)
            return xs[0, 0].double(), xs[:, 0]
        return mean_log_prob(flow, x, acc=acc)

    t_prep0 = time.perf_counter()
    mean, lp = step()                                        # includes the one-off parameter prep
    torch.cuda.synchronize()
    prep_s = time.perf_counter() - t_prep0
    for _ in range(max(args.warmup - 1, 0)):
        step()
    if not args.no_kernel_timing and args.mode != "train":
        eng.op_timing = []
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mean, lp = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    timing, eng.op_timing = eng.op_timing, None
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (per-launch HIP-event durations from the timed region) ----
    roofline = None
    if timing:
        classes = {}
        for tag, e0, e1 in timing:
            classes.setdefault(tag, []).append(e0.elapsed_time(e1))
        tot = {k: sum(v) for k, v in classes.items()}
        dom = max(tot, key=tot.get)
        avg_ms = tot[dom] / len(classes[dom])
        hs = [((h + 3) // 4) * 4 for h in args.hidden]
        peak, peak_note = F32_MFMA_PEAK_TFLOPS, "dense f32 MFMA (v_mfma_f32_32x32x2_f32)"
        if dom[0] in ("linear", "linear_bf16x3"):
            _, M, N, K = dom
            if N >= D and K >= D:                        # the D x D affine layer
                flops = 2.0 * M * D * D
                name = "linear_kernel<2,5,4,16> (BlockAffineTransform D x D)"
            else:
                flops = 2.0 * M * min(N, D) * min(K, D)
                name = f"linear_kernel (conditioner layer N={N} K={K})"
            if dom[0] == "linear_bf16x3":
                # fp32-equivalent GEMM on the bf16 matrix cores: 6 bf16 MFMA products per fp32 product, so the
                # roof for ALGORITHMIC (fp32) flops is the dense bf16 peak / 6
                name = name.replace("linear_kernel<2,5,4,16>", "linear_bf16x3_kernel<5,8,4>")
                peak, peak_note = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1), "dense bf16 MFMA peak (2500) / 6 products per fp32 product"
        else:
            _, M, ntr, npass = dom
            flops = 2.0 * M * (npass * hs[0] + sum(a * b for a, b in zip(hs[:-1], hs[1:])) + hs[-1] * ntr)
            name = "coupling_kernel (fused additive coupling)"
        ach = flops / (avg_ms * 1e-3) / 1e12
        share = tot[dom] / sum(tot.values())
        # HBM bytes per launch of that kernel: PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 runs of this
        # same command) committed under profiles/; null when the profile does not cover this kernel/shape
        traffic = None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
            base = {"linear": "linear_kernel", "linear_bf16x3": "linear_bf16x3_kernel"}.get(
                dom[0], "coupling_bf16x3_kernel" if eng.gemm_mode == "bf16x3" else "coupling_kernel")
            cand = [(v.get("dispatches", 0), v) for k, v in prof["kernels"].items() if k.split("<")[0] == base]
            if cand and B == 65536 and D == 784 and list(args.hidden) == [256, 256]:
                traffic = max(cand, key=lambda c: c[0])[1]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        # matrix-pipe utilisation of that kernel from the SQ counters (own PMC pass, profiles/r01_mfma_util.json)
        mfma_util = None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_mfma_util.json")))
            cand = [(v.get("dispatches", 0), v) for k, v in prof["kernels"].items() if k.split("<")[0] == base]
            if cand and B == 65536 and D == 784 and list(args.hidden) == [256, 256]:
                mfma_util = max(cand, key=lambda c: c[0])[1]["mfma_util"]
        except Exception:
            mfma_util = None
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "mfma_pipe_util": mfma_util, "kernel": name,
                    "peak_is": peak_note,
                    "avg_launch_ms": round(avg_ms, 4), "launches": len(classes[dom]),
                    "share_of_gpu_time": round(share, 3),
                    "algorithmic_flops_per_launch": flops,
                    "all_kernels_ms_per_step": {f"{k[0]}:{k[2]}x{k[3]}": round(v / args.steps, 3) for k, v in tot.items()}}
    # whole-flow algorithmic FLOP rate (mask-aware count)
    hs = list(args.hidden)
    n_pass = D - D // 2 if False else D // 2
    flop_per_sample = (args.blocks * (2 if args.conj else 1) + 1) * 2.0 * D * D + args.blocks * 2.0 * (
        (D // 2) * hs[0] + sum(a * b for a, b in zip(hs[:-1], hs[1:])) + hs[-1] * (D - D // 2))
    flow_tflops = flop_per_sample * value / 1e12 / world

    # ---- CPU baseline: the oracle (op-for-op torch-CPU restatement of the reference), bounded sample ----
    cpu = None
    if world == 1 and not args.no_cpu_baseline and args.mode == "log_prob":
        from oracle import usflows_oracle as orc      # the CPU oracle: this leg only
        rows = min(args.cpu_rows, B)
        xc = x[:rows].cpu()
        # 16 threads is the fastest setting for this workload on the GPU box's 2 x EPYC 9575F (probed with
        # tools/cpu_threads_probe.py: 8 -> 1860, 16 -> 2403, 32 -> 1362, 64 -> 651, 128 (torch default) -> 203 samples/s)
        torch.set_num_threads(min(args.cpu_threads, os.cpu_count() or 1))
        with torch.no_grad():
            orc.flow_log_prob(sd, spec, xc[: min(64, rows)])        # warm-up (thread pool, allocator)
            n_it, t_cpu0 = 0, time.perf_counter()
            while True:
                ref = orc.flow_log_prob(sd, spec, xc)
                n_it += 1
                if time.perf_counter() - t_cpu0 > args.cpu_seconds or n_it >= 50:
                    break
            cpu_s = time.perf_counter() - t_cpu0
        rel = ((lp[:rows].cpu().double() - ref.double()).abs() / ref.double().abs()).max().item()
        cpu = {"value": round(rows * n_it / cpu_s, 1), "unit": "samples/s", "cores": torch.get_num_threads(),
               "kind": "port",
               "sample": f"{n_it} x log_prob of the first {rows} of the {B} rows (same model/params; includes the "
                         f"per-call parameter prep the reference performs), {cpu_s:.1f} s",
               "host_cpus": os.cpu_count(), "parity_max_rel_vs_cpu_fp32": rel}

    if args.mode != "log_prob":
        cpu = None
    # warm re-prep: parameters changed (optimiser step), same storage -> the pack is refreshed in place (N1)
    prep_warm_ms = None
    if args.mode == "log_prob" and world == 1:
        with torch.no_grad():
            for p_ in flow.parameters():
                p_.add_(0.0)                       # bumps the version counters: every prepared matrix is stale
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        eng.pack(dev)
        torch.cuda.synchronize()
        prep_warm_ms = (time.perf_counter() - t_w) * 1e3
    metric = {"log_prob": "log_prob samples/sec (whole node), 32-layer 784-dim flow, batch 65536",
              "sample": "sample() samples/sec (whole node), 32-layer 784-dim flow",
              "train": "training-step samples/sec (forward + backward + Adam), 32-layer 784-dim flow"}[args.mode]
    out = {"metric": metric,
           "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" if eng.gemm_mode == "f32" else "f32 (D x D GEMMs as bf16x3 split on the bf16 MFMA, fp32-equivalent)",
           "data": "synthetic",
           "config": {"workload": f"BASELINE cfg2: USFlow in_dims=[{D}], {args.blocks} additive coupling blocks, "
                                  f"ConditionalDenseNN{list(args.hidden)}+LeakyReLU(0.01), lu_transform=1, householder={args.householder}, "
                                  f"affine_conjugation={args.conj}, "
                                  f"Laplace(0,1) base; log_prob of {B} rows per GPU resident in HBM; conditioned "
                                  f"synthetic parameters (seed 100, alpha 0.1)",
                      "rows_per_gpu": B, "global_rows": B * world, "parallelism": f"dp{world} (batch sharded, "
                      "one RCCL all-reduce of 2 fp64 scalars per step)", "fused_coupling": not args.unfused, "gemm_mode": eng.gemm_mode},
           "flow_algorithmic_tflops_per_gpu": round(flow_tflops, 2),
           "flow_frac_of_f32_mfma_peak": round(flow_tflops / F32_MFMA_PEAK_TFLOPS, 4),
           "param_prep_first_call_s": round(prep_s, 3),
           "param_prep_warm_ms": None if prep_warm_ms is None else round(prep_warm_ms, 2),
           "mean_log_prob": float(mean.item()),
           "roofline": roofline, "cpu_baseline": cpu}
    print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
