#!/usr/bin/env python3
"""Headline benchmark: log_prob samples/sec of the BASELINE cfg2 flow (D=784, 32 additive coupling
blocks, LeakyReLU MLP conditioner [256,256], Laplace base) at batch 65536 per GPU, fp32.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8 ...                     # starts 8 rank processes itself (torch.distributed.run, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...     # or under a launcher

    --config cfg2   65536 rows per GPU, weak scaling (default; the metric's configuration)
    --config cfg3   262144 rows sharded over the ranks (32768 per GPU at 8), strong scaling
    --config cfg4   D=3072, 48 blocks, hidden [1024,1024], 32768 rows per GPU
    --config cfg5   sample() of 10^6 draws sharded over the ranks (disjoint Philox substreams) + UDL check on rank 0
    --config mnist_image   the reference's MNIST experiment model (tests/explib/mnist.yaml:44-77: in_dims [16, 7, 7],
                           ConvNet2D conditioner, 2 blocks), 65536 rows per GPU
    --config cifar_image   the reference's CIFAR experiment model (experiments/cifar/cifar.yaml:56-77: in_dims [48, 8, 8],
                           10 blocks), 16384 rows per GPU
    --config mnist_live    the LIVE MNIST configuration (experiments/mnist/mnist.yaml:44-92: 15 blocks, 3 gated layers,
                           RadialDistribution(zeros[16,7,7], p=1, LogNormal(6, .35)) base, prior_scale 1.0), 65536 rows per GPU
    --config fashion_live  experiments/fashion/fashionclasses_veriflow.yaml:55-93 (10 blocks, GammaMM x 20 radial base)
    (image configurations: --base radial|laplace, --radial-norm lognormal|gammamm, --prior-scale choose the base / prior;
     mnist_image / cifar_image default to the Laplace base of rounds 2-3, the *_live ones to their YAML's radial base)

With no flags at N = 1 the line also carries an "also" block: the other BASELINE configurations, the image models and the
training steps measured in the same process after the headline (bounded steps each), so that every number DESIGN.md quotes
has the driver's clock around it.

One "step" = one full pass of the hot path over this rank's resident batch (inputs already in HBM)
+ the scalar mean-log_prob all-reduce.  Prints ONE JSON line (rank 0) with the whole-job
throughput, the roofline of the dominant kernel (HIP-event timed inside the timed region) and a
CPU baseline (the oracle; rank 0 at N=1 only)."""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0      # same guide: dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0
TIMING_EVERY = 4                    # flat log_prob / sample: per-launch HIP events on every 4th step of the timed region

CONFIGS = {
    # name: (dim, blocks, hidden, rows, sharding)   rows: per GPU ("weak") or global ("strong")
    "cfg2": dict(dim=784, blocks=32, hidden=[256, 256], rows=65536, scaling="weak", mode="log_prob"),
    "cfg3": dict(dim=784, blocks=32, hidden=[256, 256], rows=262144, scaling="strong", mode="log_prob"),
    "cfg4": dict(dim=3072, blocks=48, hidden=[1024, 1024], rows=32768, scaling="weak", mode="log_prob"),
    "cfg5": dict(dim=784, blocks=32, hidden=[256, 256], rows=1000000, scaling="strong", mode="sample"),
    # the reference's LIVE FLAT configuration as HyperoptExperiment._trial trains it (experiments/synthetic/gaussian_mixture.yaml:
    # 40-93, the 10-D overwrite): pyro.nn.DenseNN[32, 32] + ReLU, 10 blocks, affine_conjugation, RadialDistribution(p = 1,
    # GammaMM x 20) base, prior_scale 1.0, SophiaG, batch 32
    "gm_live": dict(dim=10, blocks=10, hidden=[32, 32], rows=32, scaling="weak", mode="fit", conj=True, conditioner="DenseNN",
                    negative_slope=0.0, base=dict(base="radial", radial_p=1.0, radial_norm="gammamm"),
                    extra={"gammamm_k": 20, "prior_scale": 1.0}),
}


# image-shaped flows: the models the reference's real experiments train and evaluate
IMAGE_CONFIGS = {
    "mnist_image": dict(in_dims=[16, 7, 7], blocks=2, householder=1, rows=65536, ref="tests/explib/mnist.yaml:44-77",
                        cond=dict(c_in=16, c_hidden=32, num_layers=1, padding="same", kernel_size=3, normalize_layers=True, gating=True)),
    "cifar_image": dict(in_dims=[48, 8, 8], blocks=10, householder=0, rows=16384, ref="experiments/cifar/cifar.yaml:56-77",
                        cond=dict(c_in=48, c_hidden=32, num_layers=3, padding="same", kernel_size=3, normalize_layers=True, gating=True)),
    # the live configurations as HyperoptExperiment._trial trains them: radial base, prior_scale 1.0, full depth
    "mnist_live": dict(in_dims=[16, 7, 7], blocks=15, householder=0, rows=65536, ref="experiments/mnist/mnist.yaml:44-92",
                       base="radial", radial_norm="lognormal", prior_scale=1.0,
                       cond=dict(c_in=16, c_hidden=32, num_layers=3, padding="same", kernel_size=3, normalize_layers=True, gating=True)),
    "fashion_live": dict(in_dims=[16, 7, 7], blocks=10, householder=0, rows=65536, ref="experiments/fashion/fashionclasses_veriflow.yaml:55-93",
                         base="radial", radial_norm="gammamm", prior_scale=1.0,
                         cond=dict(c_in=16, c_hidden=32, num_layers=3, padding="same", kernel_size=3, normalize_layers=True, gating=True)),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--eager-train", action="store_true", help="image --mode train: eager optimiser steps instead of Flow.fit's replayed hipGraph")
    ap.add_argument("--config", choices=sorted(CONFIGS) + sorted(IMAGE_CONFIGS), default="cfg2", help="BASELINE.json configuration")
    ap.add_argument("--batch", type=int, default=None, help="rows per GPU (overrides the configuration's)")
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--blocks", type=int, default=None)
    ap.add_argument("--hidden", type=int, nargs="+", default=None)
    ap.add_argument("--householder", type=int, default=0, help="Householder vectors per affine block (USFlow ctor default 1)")
    ap.add_argument("--conj", action="store_true", help="affine_conjugation=True (what the reference's live configs use)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the extra measurement of the opt-in fp16x2 mode")
    ap.add_argument("--unfused", action="store_true", help="run couplings as 3 linear ops instead of the fused kernel")
    ap.add_argument("--fused-min-rows", type=int, default=None, help="batch size from which couplings take the fused kernel")
    ap.add_argument("--gemm", choices=["bf16x3", "f16x2", "f32"], default=None,
                    help="GEMM arithmetic: bf16x3 = 3-way bf16 split (24 bits/operand, 6 MFMAs per product), f16x2 = 2-way "
                         "fp16 split in the planes pipeline (22 bits/operand, 3 MFMAs per product), f32 = exact-f32 MFMA")
    ap.add_argument("--merge-affine", action="store_true",
                    help="inference: consecutive affine maps (--conj) ALWAYS as one composed D x D map (FlowEngine.merge_affine = "
                         "True; the default \"auto\" composes them when a 64-row probe finds the flow well conditioned)")
    ap.add_argument("--no-merge-affine", action="store_true", help="never compose consecutive affine maps")
    ap.add_argument("--optim", choices=["sophia", "adam"], default="sophia",
                    help="--mode train: SophiaG (the reference's Flow.fit default) or torch's Adam")
    ap.add_argument("--mode", choices=["log_prob", "sample", "train", "fit"], default=None,
                    help="sample: time Flow.sample (Philox head + forward pass); train: one optimiser step of Flow.fit's "
                         "loss (-log_prob.mean(): device forward + backward + optimiser); fit (flat configurations): the same "
                         "step as Flow.fit issues it on its own stream -- captured as a hipGraph and replayed at small batches")
    ap.add_argument("--base", choices=["laplace", "radial"], default=None,
                    help="image configurations: Laplace(0, 1) base or RadialDistribution(zeros[C,H,W], p=1, norm) -- the live YAMLs' base")
    ap.add_argument("--radial-norm", choices=["lognormal", "gammamm"], default=None,
                    help="--base radial: LogNormal(6, .35) (mnist.yaml:79-92) or GammaMM x 20 (fashionclasses_veriflow.yaml:79-93)")
    ap.add_argument("--prior-scale", type=float, default=None, help="image configurations: USFlow(prior_scale=...) (live YAMLs: 1.0)")
    ap.add_argument("--force-dist", action="store_true",
                    help="--gpus 1: initialise a one-rank \"nccl\" (= RCCL) process group in this process and issue the step's collective "
                         "anyway (the all-reduce of one rank is the identity): the RCCL path under the bench's clock on a single GPU")
    ap.add_argument("--golden", default=None,
                    help="flat log_prob: name of a committed fixture of the REAL reference (tests/golden/<name>.npz, its parameters this "
                         "generator's at the fixture's seed): its rows are placed at the head / middle / tail of the batch and the "
                         "device log_prob of them is compared with the reference's fp64 run (probe_parity)")
    ap.add_argument("--no-also", action="store_true", help="default invocation: skip the \"also\" block")
    ap.add_argument("--seed", type=int, default=100, help="seed of the synthetic parameters")
    ap.add_argument("--cpu-rows", type=int, default=4096, help="rows of the short CPU sample (second cpu_baseline figure)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--cpu-full-iters", type=int, default=3, help="full-batch iterations of the CPU baseline (median)")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--device", choices=["cuda", "cpu"], default="cuda",
                    help="cpu: PLUMBING TEST of the launcher / collective path on gloo with the torch composite "
                         "formulation -- the line says so and is not a measurement")
    return ap.parse_args(argv)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args) -> int:
    """--gpus N > 1 outside a launcher: this parent (which never touches the GPU) starts N fresh rank processes under
    torch.distributed.run and relays their output; rank 0 prints the JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    under_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # started by torch.distributed.run
    if args.gpus > 1 and not under_launcher:
        sys.exit(launch_ranks(args))
    # stdout carries exactly ONE line, the JSON: whatever a library prints to file descriptor 1 meanwhile (RCCL announces its
    # version there when a communicator is created) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        out = run_config(args, under_launcher)
        if out is not None:
            default_line = (len(sys.argv) == 1 or all(a.split("=")[0] in ("--gpus", "--steps", "--warmup") or a.lstrip("-").isdigit()
                                                     for a in sys.argv[1:]))
            if default_line and args.gpus == 1 and not under_launcher and not args.no_also and args.device == "cuda":
                out["also"] = run_also(args)
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if out is None:                 # (a rank other than 0)
        return
    print(json.dumps(out), flush=True)


ALSO = [
    # (label, argv) -- measured in this process after the headline; bounded steps; the CPU legs on small samples
    ("cfg4 log_prob (D=3072, 48 blocks, 32768 rows; golden rows of the real reference at head/middle/tail)",
     ["--config", "cfg4", "--steps", "3", "--warmup", "1", "--no-fast-mode", "--no-cpu-baseline", "--seed", "102"]),
    ("cfg5 per-rank share: sample() of 125000 draws (rank 0's Philox substream) + UDL check",
     ["--config", "cfg5", "--batch", "125000", "--steps", "3", "--warmup", "1"]),
    ("mnist_image log_prob, Laplace base (rounds 2-3 workload)",
     ["--config", "mnist_image", "--steps", "10", "--warmup", "3", "--cpu-seconds", "2", "--cpu-rows", "2048"]),
    ("mnist_live log_prob: the live MNIST configuration (15 blocks, 3 gated layers, radial LogNormal base)",
     ["--config", "mnist_live", "--steps", "5", "--warmup", "2", "--cpu-seconds", "2", "--cpu-rows", "256"]),
    ("mnist_live log_prob at 100 rows: the reference's evaluation chunk (hyperopt.py:273-278), the recorded op list replayed",
     ["--config", "mnist_live", "--batch", "100", "--steps", "50", "--warmup", "10", "--cpu-seconds", "1", "--cpu-rows", "100", "--no-kernel-timing"]),
    ("cifar_image log_prob, live base (radial LogNormal, prior_scale 1)",
     ["--config", "cifar_image", "--base", "radial", "--prior-scale", "1", "--steps", "5", "--warmup", "2", "--cpu-seconds", "2", "--cpu-rows", "128"]),
    ("cfg2 log_prob through a one-rank nccl (RCCL) group: communicator init + the scalar all-reduce on the compute stream",
     ["--config", "cfg2", "--force-dist", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-fast-mode", "--no-kernel-timing"]),
    # SURVEY section 8's two mandatory variants and section 8d's secondary base, at the headline's depth and batch, golden rows of the
    # real reference at head / middle / tail
    ("cfg2 log_prob with affine_conjugation=True, householder=0 (what every live configuration uses)",
     ["--config", "cfg2", "--conj", "--seed", "103", "--golden", "synth_d784_k32_conj", "--steps", "5", "--warmup", "2", "--no-fast-mode",
      "--no-cpu-baseline"]),
    ("cfg2 log_prob with affine_conjugation=True, householder=1 (the USFlow constructor's default, flows.py:402)",
     ["--config", "cfg2", "--conj", "--householder", "1", "--seed", "104", "--golden", "synth_d784_k32_hh1_conj", "--steps", "5", "--warmup", "2",
      "--no-fast-mode", "--no-cpu-baseline"]),
    ("cfg2 log_prob with the secondary base RadialDistribution(p=1, LogNormal(6, .35)) (tests/explib/mnist.yaml:79-92)",
     ["--config", "cfg2", "--base", "radial", "--seed", "105", "--golden", "synth_d784_k32_radial1", "--steps", "5", "--warmup", "2", "--no-fast-mode",
      "--no-cpu-baseline"]),
    ("cfg2 training step at 65536 rows (forward + backward + SophiaG)",
     ["--config", "cfg2", "--mode", "train", "--steps", "5", "--warmup", "2"]),
    ("cfg2 Flow.fit step at batch 32 (the reference's training batch), replayed hipGraph",
     ["--config", "cfg2", "--mode", "fit", "--batch", "32", "--steps", "30", "--warmup", "6"]),
    ("gm_live Flow.fit step at batch 32: the live FLAT configuration (gaussian_mixture.yaml:40-93, 10-D: DenseNN[32,32]+ReLU, 10 blocks, "
     "conj, GammaMM x 20 radial base, prior_scale 1, SophiaG), replayed hipGraph",
     ["--config", "gm_live", "--steps", "30", "--warmup", "6"]),
    ("mnist_image training step at 65536 rows, live base (radial LogNormal, prior_scale 1)",
     ["--config", "mnist_image", "--mode", "train", "--base", "radial", "--prior-scale", "1", "--steps", "5", "--warmup", "5",
      "--cpu-seconds", "1", "--cpu-rows", "512"]),
    ("mnist_image Flow.fit step at batch 32, live base, replayed hipGraph",
     ["--config", "mnist_image", "--mode", "train", "--base", "radial", "--prior-scale", "1", "--batch", "32", "--steps", "50", "--warmup", "6",
      "--cpu-seconds", "1"]),
    ("mnist_live Flow.fit step at batch 32: what HyperoptExperiment._trial runs (mnist.yaml:30-92), replayed hipGraph",
     ["--config", "mnist_live", "--mode", "train", "--batch", "32", "--steps", "30", "--warmup", "6", "--cpu-seconds", "1"]),
]


_TRAIN_WGRADS = ("usf_wgrad_f32", "usf_wgrad_bias_f32", "usf_wgrad_planes_f32", "usf_wgrad_blocked_f32")
_TRAIN_TIMED = _TRAIN_WGRADS + ("usf_gemm_planes_bf16x3", "usf_coupling_planes", "usf_coupling_additive_f32", "usf_linear_f32",
                                "usf_pack_planes_f32", "usf_run_ops", "usf_gemm_f64", "usf_pack_weights_f32", "usf_pack_weights_t_f32")
ALSO_BUDGET_S = 150.0            # (measured: 37 s for all entries on an idle box)


def run_also(args):
    """the other workloads DESIGN.md quotes, each measured by run_config in this process (same contract: W warm-up steps,
    K timed steps, device-synchronised), condensed to value / ms_per_step / dominant-kernel roofline / parity"""
    import gc
    out = []
    t_all = time.perf_counter()
    for label, argv in ALSO:
        t0 = time.perf_counter()
        rec = {"workload": label, "argv": " ".join(argv)}
        if t0 - t_all > ALSO_BUDGET_S:
            # (a slow box: the default command must stay within a few minutes whatever happens; the entry can be run by hand)
            rec["skipped"] = f"time budget of the also block ({ALSO_BUDGET_S} s) spent"
            out.append(rec)
            continue
        try:
            a = parse_args(argv)
            if a.config == "cfg4":
                a.probe_rows = _golden_probe("synth_d3072_k48_cfg4", a.seed)
            elif a.golden:
                a.probe_rows = _golden_probe(a.golden, a.seed)
            o = run_config(a, False)
            rl = o.get("roofline") or {}
            cb = o.get("cpu_baseline") or {}
            parity = {}
            if cb.get("parity_max_rel_vs_cpu_fp32") is not None:
                parity["max_rel_vs_cpu_fp32_oracle"] = cb["parity_max_rel_vs_cpu_fp32"]
                parity["rows"] = cb.get("parity_rows")
            for k in ("probe_parity", "train_parity", "udl_check"):
                if o.get(k):
                    parity[k] = o[k]
            rec.update({"metric": o["metric"], "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"],
                        "steps": o["steps"], "warmup": o["warmup"], "rows_per_gpu": o["config"]["rows_per_gpu"],
                        "roofline": {k: rl.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                                            "share_of_gpu_time", "kernel_ms_per_step", "launch_bound")} if rl else None,
                        "parity": parity or None, "train_step": o.get("train_step"),
                        "cpu_baseline": ({k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")} if cb else None),
                        "mean_log_prob": o.get("mean_log_prob"), "backend": o.get("backend"),
                        "parallelism": o["config"].get("parallelism")})
        except Exception as e:      # noqa: BLE001 -- one failing extra must not cost the headline its line
            rec["error"] = f"{type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}"
        rec["wall_s"] = round(time.perf_counter() - t0, 1)
        out.append(rec)
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    return {"what": "measured in this process after the headline (bench.py run_also); every entry: W untimed warm-up steps, K timed "
                    "steps between device synchronisations, inputs resident in HBM",
            "wall_s": round(time.perf_counter() - t_all, 1), "entries": out}


def _golden_probe(name, seed):
    """(x rows, the real reference's fp64 log_prob of them) of a committed golden fixture whose parameters are this
    generator's at `seed` (tests/golden/make_golden.py); None when the fixture is absent"""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    if not os.path.isfile(path):
        return None
    z = np.load(path, allow_pickle=False)
    if int(z["seed"]) != int(seed) or abs(float(z["alpha"]) - 0.1) > 1e-12:
        return None
    return torch.from_numpy(z["x"]), torch.from_numpy(z["log_prob64"])


def run_config(args, under_launcher):
    """one configuration measured in this process; the JSON line's dict on rank 0, None on the other ranks"""
    if args.config in IMAGE_CONFIGS:
        return main_image(args, under_launcher)
    return main_flat(args, under_launcher)


def main_flat(args, under_launcher):
    cfg = CONFIGS[args.config]
    mode = args.mode or cfg["mode"]
    D = args.dim or cfg["dim"]
    blocks = args.blocks or cfg["blocks"]
    hidden = list(args.hidden or cfg["hidden"])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1
    on_gpu = args.device == "cuda"
    import torch.distributed as dist
    backend = None
    own_group = False
    if under_launcher:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if on_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo")
        backend = dist.get_backend()
    elif args.force_dist:
        # one rank, its own rendezvous: the collective path (communicator init, all-reduce on the compute stream) on ONE GPU
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if on_gpu:
            torch.cuda.set_device(0)
        dist.init_process_group("nccl" if on_gpu else "gloo", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1,
                                **({"device_id": torch.device("cuda:0")} if on_gpu else {}))
        backend = dist.get_backend()
        own_group = under_launcher = True               # (barriers, the max over ranks and the teardown below run as under a launcher)
    dev = torch.device(f"cuda:{local_rank}") if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(dev)
    sync = (lambda: torch.cuda.synchronize()) if on_gpu else (lambda: None)

    from usflows_amd.synth import ModelSpec, synth_state_dict, build_usflow
    from usflows_amd.parallel import mean_log_prob, sample_sharded, shard_rows

    if getattr(args, "probe_rows", None) is None and getattr(args, "golden", None):
        args.probe_rows = _golden_probe(args.golden, args.seed)
    # --base radial: the secondary base of SURVEY section 8d on the flat model -- RadialDistribution(p = 1, LogNormal(6, .35))
    # (tests/explib/mnist.yaml:79-92); density on usf_radial_logprob_f32
    base_kw = dict(base="laplace")
    if args.base == "radial":
        base_kw = dict(base="radial", radial_p=1.0, radial_norm="lognormal", radial_norm_loc=6.0, radial_norm_scale=0.35)
    if cfg.get("base"):
        base_kw = dict(cfg["base"])
    if cfg.get("conj"):
        args.conj = True
    spec = ModelSpec(D, blocks, hidden, householder=args.householder, affine_conjugation=args.conj,
                     negative_slope=cfg.get("negative_slope", 0.01), conditioner=cfg.get("conditioner", "ConditionalDenseNN"),
                     extra=dict(cfg.get("extra", {})), **base_kw)
    sd = synth_state_dict(spec, seed=args.seed, alpha=0.1)   # same parameters on every rank
    flow = build_usflow(spec, sd, device=str(dev))
    eng = flow.engine() if on_gpu else None
    if on_gpu:
        if eng is None:
            raise RuntimeError("bench.py: the flow has no device engine -- refusing to time a fallback")
        eng.use_fused_coupling = not args.unfused
        if args.merge_affine:
            eng.merge_affine = True      # consecutive affine maps (affine_conjugation) composed at pack time, unconditionally
        if args.no_merge_affine:
            eng.merge_affine = False
        if args.fused_min_rows is not None:
            eng.fused_min_rows = args.fused_min_rows
        if args.gemm:
            eng.gemm_mode = args.gemm

    # ---- this rank's share of the workload ----
    scaling = cfg["scaling"] if args.batch is None else "weak"
    if scaling == "weak":
        B = args.batch or cfg["rows"]                       # per GPU
        global_rows, lo = B * world, rank * B
    else:
        global_rows = cfg["rows"]
        lo, hi = shard_rows(global_rows, rank, world)       # contiguous row range of the global batch
        B = hi - lo
    x = None
    if mode != "sample":
        g = torch.Generator().manual_seed(1234 + rank)
        x = torch.rand(B, D, generator=g).to(dev)           # this rank's shard, resident in HBM
    acc = torch.zeros(2, dtype=torch.float64, device=dev)

    # rows of the REAL reference's golden run placed at the head / middle / tail of the batch (the fixture's parameters are
    # this generator's at the fixture's seed): parity of configurations whose oracle call would take minutes (cfg4)
    probe = None
    if getattr(args, "probe_rows", None) is not None and x is not None:
        px, plp = args.probe_rows
        n = px.shape[0]
        a_, b_ = n // 3, 2 * (n // 3)
        pidx = torch.cat([torch.arange(0, a_), torch.arange(B // 2 - (b_ - a_) // 2, B // 2 - (b_ - a_) // 2 + (b_ - a_)),
                          torch.arange(B - (n - b_), B)])
        x[pidx.to(dev)] = px.to(dev)
        probe = (pidx.to(dev), plp)

    opt = None
    if mode in ("train", "fit"):
        # the optimiser Flow.fit defaults to (flows.py:116): SophiaG -- one multi-tensor HIP launch per step on the GPU
        from usflows_amd.sophia import SophiaG
        opt = SophiaG(flow.parameters(), lr=1e-6) if args.optim == "sophia" else torch.optim.Adam(flow.parameters(), lr=1e-6)
    if mode in ("train", "fit") and under_launcher and (world > 1 or args.force_dist):
        from usflows_amd.parallel import data_parallel_training
        data_parallel_training(flow, force_collective=args.force_dist)      # one all-reduce of the flat gradient arena per step
    fit_replays = [0]

    def fit_step():
        # what Flow._fit_epochs does per batch (usflows_amd/flows.py): the replayed graph when there is one, else an eager step
        loss = flow._train_graph_step(opt, x, None) if (world == 1 and not args.force_dist) else None
        if loss is not None:
            fit_replays[0] += 1
            return torch.tensor(-loss, dtype=torch.float64), None
        flow._zero_grad_for_step(opt)
        lp_ = flow.log_prob(x)
        loss = -lp_.mean() - flow.log_prior()
        loss.backward()
        opt.step()
        return -loss.detach().double(), lp_.detach()

    def step():
        if mode == "fit":
            with flow.fit_stream(dev):                      # Flow.fit runs its loop on a stream of the flow's own
                return fit_step()
        if mode == "train":
            opt.zero_grad(set_to_none=True)
            lp_ = flow.log_prob(x)
            loss = -lp_.mean()
            loss.backward()
            opt.step()
            return -loss.detach().double(), lp_.detach()
        if mode == "sample":
            with torch.no_grad():
                xs = flow.sample([B], seed=1234, row_offset=lo)       # rows [lo, lo+B) of the global draw
            return xs[0, 0].double(), xs
        return mean_log_prob(flow, x, acc=acc, force_collective=args.force_dist)   # + ONE all-reduce of [sum log_prob, count] when world > 1

    t_prep0 = time.perf_counter()
    mean, lp = step()                                        # includes the one-off parameter prep
    sync()
    prep_s = time.perf_counter() - t_prep0
    for _ in range(max(args.warmup - 1, 4 if mode == "fit" else 0)):
        step()
    fit_replays[0] = 0
    # per-launch HIP events: on every TIMING_EVERY-th step of the timed region (first step included).  Two event records per
    # launch on all 66 launches of every step cost the cfg2 step 2 % (18.55 vs 18.18 ms, same box): the headline would measure
    # its own instrumentation; sampled, the events cost 0.5 % and still see >= 160 launches of the dominant kernel
    timing_on = on_gpu and not args.no_kernel_timing and mode not in ("train", "fit")
    timed_launch_steps = [0]
    op_records = []
    if on_gpu and not args.no_kernel_timing and mode == "train":
        from usflows_amd import _ext as _ext_t
        # HIP events around the launches of the training step's kernel classes -- on every TIMING_EVERY-th step only (two event records
        # around each of the ~430 launches cost the 68 ms step 3.5 ms)
        train_events = {n_: [] for n_ in _TRAIN_TIMED}
        train_timed_steps = 0
    if under_launcher:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i_step in range(args.steps):
        if timing_on and i_step % TIMING_EVERY == 0:
            eng.op_timing = op_records
            timed_launch_steps[0] += 1
        if on_gpu and not args.no_kernel_timing and mode == "train":
            _ext_t.launch_timing = train_events if i_step % TIMING_EVERY == 0 else None
            train_timed_steps += 1 if i_step % TIMING_EVERY == 0 else 0
        mean, lp = step()
        if timing_on:
            eng.op_timing = None
    sync()
    if under_launcher:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    timing = None
    wg_timing = None
    if on_gpu:
        timing = op_records if timing_on else None
        eng.op_timing = None
        if mode == "train" and not args.no_kernel_timing:
            from usflows_amd import _ext as _ext_t
            _ext_t.launch_timing = None
            wg_timing = [(e0, e1, a, fn) for fn in _TRAIN_WGRADS for e0, e1, a in train_events[fn]]
            train_classes = {fn: (len(v), sum(e0.elapsed_time(e1) for e0, e1, _a in v)) for fn, v in train_events.items() if v}
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if under_launcher:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = global_rows * args.steps / elapsed

    # ---- opt-in fast mode, measured beside the headline (same process, same inputs, every rank): the planes pipeline
    # with fp16x2 planes -- 22 significant bits per operand instead of 24, three MFMAs per product instead of six.
    # NOT the headline: `value` above is the default (bf16x3) mode.
    fast = None
    if on_gpu and mode == "log_prob" and args.gemm is None and not args.no_fast_mode:
        lp_default = lp.clone()
        eng.gemm_mode = "f16x2"
        for _ in range(3):
            mean_f, lp_f = step()
        if under_launcher:
            dist.barrier()
        sync()
        tf0 = time.perf_counter()
        for _ in range(args.steps):
            mean_f, lp_f = step()
        sync()
        if under_launcher:
            dist.barrier()
        tfe = torch.tensor([time.perf_counter() - tf0], dtype=torch.float64, device=dev)
        if under_launcher:
            dist.all_reduce(tfe, op=dist.ReduceOp.MAX)
        el_f = float(tfe.item())
        lp_fast = lp_f.clone()
        fast = {"gemm_mode": "f16x2", "value": round(global_rows * args.steps / el_f, 1), "unit": "samples/s",
                "ms_per_step": round(el_f / args.steps * 1e3, 3),
                "max_rel_vs_default_mode": float(((lp_fast.double() - lp_default.double()).abs() / lp_default.double().abs()).max().item()),
                "range_guard_fallbacks": eng.f16_fallbacks,
                "what": "planes pipeline: activations travel between layers as two fp16 planes in MFMA-operand order, "
                        "a1 w1 + a1 w2 + a2 w1 on v_mfma_f32_16x16x32_f16, fp32 accumulation; engine.gemm_mode = 'f16x2' "
                        "or USFLOWS_AMD_GEMM=f16x2; same batch, parameters and step definition as the headline"}
        eng.gemm_mode = "bf16x3"

    if rank != 0:
        if under_launcher:
            dist.destroy_process_group()
        return None

    # ---- roofline of the dominant kernel (per-launch HIP-event durations from the timed region) ----
    roofline = None
    if timing:
        classes = {}
        for tag, e0, e1 in timing:
            classes.setdefault(tag, []).append(e0.elapsed_time(e1))
        tot = {k: sum(v) for k, v in classes.items()}
        dom = max(tot, key=tot.get)
        avg_ms = tot[dom] / len(classes[dom])
        hs = [((h + 3) // 4) * 4 for h in hidden]
        peak, peak_note = F32_MFMA_PEAK_TFLOPS, "dense f32 MFMA (v_mfma_f32_32x32x2_f32)"
        lib_variant = None
        if dom[0] in ("linear", "linear_bf16x3"):
            _, M, N, K = dom
            if N >= D and K >= D:                        # the D x D affine layer
                flops = 2.0 * M * D * D
                what = "BlockAffineTransform D x D"
            else:
                flops = 2.0 * M * min(N, D) * min(K, D)
                what = f"conditioner layer N={N} K={K}"
            lib_variant = _variant_code(M, N, K, dom[0] == "linear_bf16x3")
            name = f"{_variant_name(lib_variant)} ({what})"
            base = "linear_bf16x3_kernel" if dom[0] == "linear_bf16x3" else "linear_kernel"
            if dom[0] == "linear_bf16x3":
                # fp32-equivalent GEMM on the bf16 matrix cores: 6 bf16 MFMA products per fp32 product, so the
                # roof for ALGORITHMIC (fp32) flops is the dense bf16 peak / 6
                peak, peak_note = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1), "dense bf16 MFMA peak (2500) / 6 products per fp32 product"
        elif dom[0] == "gemm_planes":
            _, M, N, K = dom
            flops = 2.0 * M * min(N, D) * min(K, D) if (N >= D and K >= D) else 2.0 * M * N * K
            what = "BlockAffineTransform D x D" if (N >= D and K >= D) else f"conditioner layer N={N} K={K}"
            f16 = eng.gemm_mode == "f16x2"
            name = f"gemm_planes_kernel<{2 if f16 else 3},5,planes> ({what})"
            base = "gemm_planes_kernel"
            if f16:
                peak, peak_note = round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1), "dense f16 MFMA peak (2500) / 3 products per fp32 product (fp16x2 split)"
            else:
                peak, peak_note = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1), "dense bf16 MFMA peak (2500) / 6 products per fp32 product"
        else:
            _, M, ntr, npass = dom
            flops = 2.0 * M * (npass * hs[0] + sum(a * b for a, b in zip(hs[:-1], hs[1:])) + hs[-1] * ntr)
            name = "coupling_bf16x3_kernel (fused additive coupling)" if eng.gemm_mode == "bf16x3" else \
                "coupling_kernel (fused additive coupling)"
            base = "coupling_bf16x3_kernel" if eng.gemm_mode == "bf16x3" else "coupling_kernel"
            if eng.gemm_mode == "bf16x3":
                peak, peak_note = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1), "dense bf16 MFMA peak (2500) / 6 products per fp32 product"
        ach = flops / (avg_ms * 1e-3) / 1e12
        share = tot[dom] / sum(tot.values())
        # HBM bytes per launch / matrix-pipe utilisation of that kernel: NOT measured by this process -- PMC counters need
        # rocprofv3 around the run.  They are copied from the committed summaries of separate `rocprofv3 --pmc` passes
        # of this same command (tools/make_profiles.py), and the line says so; null when no summary covers this shape
        traffic, traffic_src = _from_profile("hbm_traffic", base, "hbm_bytes_per_launch", args, B, D, hidden)
        mfma_util, util_src = _from_profile("mfma_util", base, "mfma_util", args, B, D, hidden)
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "mfma_pipe_util": mfma_util, "mfma_pipe_util_source": util_src, "kernel": name,
                    "peak_is": peak_note,
                    "measured_by": f"HIP events around every launch of every {TIMING_EVERY}th step of this run's timed region "
                                   f"(events on all steps cost the step 2 %: bench.py TIMING_EVERY)",
                    "avg_launch_ms": round(avg_ms, 4), "launches": len(classes[dom]),
                    "share_of_gpu_time": round(share, 3),
                    "algorithmic_flops_per_launch": flops,
                    "all_kernels_ms_per_step": {f"{k[0]}:{k[2]}x{k[3]}": round(v / max(timed_launch_steps[0], 1), 3) for k, v in tot.items()},
                    "instrumented_steps": timed_launch_steps[0]}
    if wg_timing:
        # training step: the weight gradients are the largest kernel class (usf_wgrad_f32 = the wgrad kernel + the
        # reduction of its row-range partials); the D x D launches of the affine layers are the dominant shape
        from usflows_amd import _ext as _ext_t
        shapes = {}
        for e0, e1, a, fn in wg_timing:
            mnk = ((int(a[8]), int(a[9]), int(a[10])) if fn == "usf_wgrad_planes_f32" else
                   (int(a[6]), int(a[7]), int(a[8])) if fn == "usf_wgrad_blocked_f32" else (int(a[4]), int(a[5]), int(a[6])))
            shapes.setdefault(("usf_wgrad_f32" if fn == "usf_wgrad_bias_f32" else fn,) + mnk, []).append(e0.elapsed_time(e1))
        tot = {k: sum(v) for k, v in shapes.items()}
        dom = max(tot, key=tot.get)
        fn_, M_, N_, K_ = dom
        avg_ms = tot[dom] / len(shapes[dom])
        flops = 2.0 * M_ * N_ * K_
        if fn_ == "usf_wgrad_blocked_f32":
            bf = True
            kname = ("wgrad_planes_kernel<blocked> (bf16x3, both operands the planes buffers of the planes pipeline; one block per CU) + "
                     f"reduce_partials_cls_kernel: usf_wgrad_blocked_f32 N={N_} K={K_} over {M_} rows")
        elif fn_ == "usf_wgrad_planes_f32":
            bf = True
            kname = ("wgrad_planes_kernel (bf16x3 from the operand planes the layer's GEMMs wrote; one block per CU) + "
                     f"reduce_partials_cls_kernel: usf_wgrad_planes_f32 N={N_} K={K_} over {M_} rows")
        else:
            variant = _ext_t.load().usf_wgrad_variant(M_, N_, K_, (N_ + 3) // 4 * 4, (K_ + 3) // 4 * 4, 1 if eng.gemm_mode != "f32" else 0)
            bf = variant != 0
            kname = ({0: "wgrad_kernel (exact f32)", 1: "wgrad_bf16x3_kernel", 2: "wgrad_lw_kernel (bf16x3, loader waves)"}[variant]
                     + f" + reduce_partials_kernel: usf_wgrad_f32 N={N_} K={K_} over {M_} rows")
        peak = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1) if bf else F32_MFMA_PEAK_TFLOPS
        ach = flops / (avg_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": None, "traffic_source": None,
                    "kernel": kname,
                    "peak_is": "dense bf16 MFMA peak (2500) / 6 products per fp32 product" if bf else "dense f32 MFMA",
                    "measured_by": f"HIP events around every {fn_} launch of every {TIMING_EVERY}th step of this run's timed region",
                    "avg_launch_ms": round(avg_ms, 4), "launches": len(shapes[dom]),
                    "algorithmic_flops_per_launch": flops,
                    "wgrad_ms_per_step": {f"{k[2]}x{k[3]}" + (" (planes)" if k[0] == "usf_wgrad_planes_f32" else
                                                               " (blocked planes)" if k[0] == "usf_wgrad_blocked_f32" else ""): round(v / max(train_timed_steps, 1), 3)
                                          for k, v in tot.items()},
                    "instrumented_steps": train_timed_steps,
                    "wgrad_frac_by_shape": {f"{k[2]}x{k[3]}": round(2.0 * k[1] * k[2] * k[3] / (v / len(shapes[k]) * 1e-3) / 1e12 / peak, 3)
                                            for k, v in tot.items()},
                    "launch_classes_ms_per_step": {fn: round(ms_ / max(train_timed_steps, 1), 3) for fn, (_n, ms_) in train_classes.items()},
                    "wgrad_share_of_step": round(sum(tot.values()) / max(train_timed_steps, 1) / ms_per_step, 3)}
    if mode == "fit" and roofline is not None:
        #  Flow.fit at the reference's batch is a chain of dependent launches: no kernel roofline applies
        roofline.update({"bound": "launch", "frac": None, "achieved": None, "peak": None, "unit": None,
                         "peak_is": "no kernel roofline applies: the replayed step is bound by the number of dependent launches"})
    if roofline is not None and on_gpu and roofline.get("bound") == "mfma" and "bf16" in str(roofline.get("peak_is", "")):
        # the ceiling this part SUSTAINS for the kernels' matrix-core instruction mix, measured in THIS run right behind the timed
        # region (usf_mfma_probe: a register-only loop of v_mfma_f32_16x16x32_bf16 at the GEMM's occupancy and tiling; under
        # dense MFMA load the chip is power-bound and its clock sits below the 2.4 GHz the nominal 2.5 PFLOP/s assume)
        from usflows_amd import _ext as _ext_p
        pr = _ext_p.mfma_probe(dev, iters=400, repeats=5)
        roofline["sustained_peak"] = round(pr["tflops_f32_equiv"], 1)
        roofline["frac_of_sustained"] = round(roofline["achieved"] / pr["tflops_f32_equiv"], 4)
        roofline["sustained_peak_is"] = (f"usf_mfma_probe in this run: register-only loop of the planes GEMM's MFMA mix, {pr['tflops_bf16']:.0f} TFLOP/s "
                                         f"bf16 = {pr['tflops_bf16'] / BF16_MFMA_PEAK_TFLOPS:.2f} of nominal, / 6 products per fp32 product; "
                                         f"median of {pr['launches']} launches of {pr['ms']:.2f} ms")
        if mode == "log_prob":
            # the clock the matrix cores actually run at (usf_set_clock_buffer: every block's lifetime in shader cycles over the same
            # in ticks of the constant 100 MHz counter) under the step's planes GEMMs and under the probe: the nominal peak assumes
            # 2400 MHz, the part holds less under dense MFMA load -- `frac_at_own_clock` prices the GEMM against the roof AT THE CLOCK
            # IT RAN AT (what is left is issue slots the kernel does not fill; the clock itself is what its data movement costs)
            with _ext_p.clock_meter(dev) as cm:
                for _ in range(3):
                    step()
                sync()
            with _ext_p.clock_meter(dev) as cm2:
                _ext_p.mfma_probe(dev, iters=400, repeats=3)
            g_mhz, p_mhz = cm.mhz(), cm2.mhz()
            if g_mhz and p_mhz:
                roofline["clock_mhz"] = {"under_gemm_planes": round(g_mhz), "under_mfma_probe": round(p_mhz), "nominal": 2400,
                                         "how": "s_memtime / s_memrealtime over every block's lifetime, 3 steps / 4 probe launches behind the timed region"}
                roofline["frac_at_own_clock"] = round(roofline["achieved"] / (roofline["peak"] * g_mhz / 2400.0), 4)
    # whole-flow algorithmic FLOP rate (mask-aware count)
    hs = list(hidden)
    flop_per_sample = (blocks * (2 if args.conj else 1) + 1) * 2.0 * D * D + blocks * 2.0 * (
        (D // 2) * hs[0] + sum(a * b for a, b in zip(hs[:-1], hs[1:])) + hs[-1] * (D - D // 2))
    flow_tflops = flop_per_sample * value / 1e12 / world

    # ---- cfg5: UDL-preservation check of the drawn samples (rank 0, outside the timed region) ----
    udl = None
    if mode == "sample" and on_gpu:
        with torch.no_grad():
            xs = lp[: min(B, 65536)]
            lpx = flow.log_prob(xs)
            zb = flow.backward(xs)
        base_lp = torch.distributions.Laplace(0.0, 1.0).log_prob(zb.double()).sum(-1)
        const = lpx.double() - base_lp
        udl = {"rows_checked": int(xs.shape[0]), "constant": float(const.mean().item()),
               "max_abs_dev_rel": float(((const - const.mean()).abs().max() / base_lp.abs().max()).item()),
               "passes_1e-5": bool(((const - const.mean()).abs().max() / base_lp.abs().max()).item() < 1e-5),
               "all_finite": bool(torch.isfinite(xs).all().item())}

    # ---- CPU baseline: the oracle (op-for-op torch-CPU restatement of the reference), rank 0 at N = 1 only ----
    cpu = None
    # (on the metric's configuration only: one oracle call at cfg4 re-inverts 98 triangular 3072 x 3072 factors, minutes)
    if world == 1 and not args.no_cpu_baseline and mode == "log_prob" and on_gpu and args.config == "cfg2" and B <= 65536:
        from oracle import usflows_oracle as orc      # the CPU oracle: this leg only
        # 16 threads is the fastest setting for this workload on the GPU box's 2 x EPYC 9575F (probed with
        # tools/cpu_threads_probe.py: 8 -> 1860, 16 -> 2403, 32 -> 1362, 64 -> 651, 128 (torch default) -> 203 samples/s)
        torch.set_num_threads(min(args.cpu_threads, os.cpu_count() or 1))
        xc = x.cpu()
        rows = min(args.cpu_rows, B)
        with torch.no_grad():
            orc.flow_log_prob(sd, spec, xc[: min(64, rows)])        # warm-up (thread pool, allocator)
            # (1) the metric's own batch: full B rows, >= 3 iterations, median
            full_t, ref = [], None
            for _ in range(max(args.cpu_full_iters, 0)):
                t_c = time.perf_counter()
                ref = orc.flow_log_prob(sd, spec, xc)
                full_t.append(time.perf_counter() - t_c)
            # (2) a short sample (first `rows` rows), where the per-call parameter prep is about half of each call
            n_it, t_cpu0 = 0, time.perf_counter()
            while True:
                ref_s = orc.flow_log_prob(sd, spec, xc[:rows])
                n_it += 1
                if time.perf_counter() - t_cpu0 > args.cpu_seconds or n_it >= 50:
                    break
            cpu_s = time.perf_counter() - t_cpu0
        if ref is not None:
            rel = ((lp.cpu().double() - ref.double()).abs() / ref.double().abs())
            sample = (f"{len(full_t)} x log_prob of all {B} rows of the batch (same model/params; includes the per-call "
                      f"parameter prep the reference performs), median {statistics.median(full_t):.1f} s per call")
            v_full = B / statistics.median(full_t)
        else:
            rel = ((lp[:rows].cpu().double() - ref_s.double()).abs() / ref_s.double().abs())
            sample, v_full = f"{n_it} x log_prob of the first {rows} rows", rows * n_it / cpu_s
        if fast is not None and ref is not None:
            fast["parity_max_rel_vs_cpu_fp32"] = float(((lp_fast.cpu().double() - ref.double()).abs() / ref.double().abs()).max().item())
        third = max(B // 3, 1)
        cpu = {"value": round(v_full, 1), "unit": "samples/s", "cores": torch.get_num_threads(),
               "kind": "port", "sample": sample, "host_cpus": os.cpu_count(),
               "value_short_sample": round(rows * n_it / cpu_s, 1),
               "short_sample": f"{n_it} x log_prob of the first {rows} rows, {cpu_s:.1f} s",
               "parity_rows": int(rel.numel()),
               "parity_max_rel_vs_cpu_fp32": float(rel.max().item()),
               "parity_max_rel_head_middle_tail": [float(rel[:third].max().item()), float(rel[third: 2 * third].max().item()),
                                                   float(rel[2 * third:].max().item())]}

    # warm re-prep: parameters changed (optimiser step), same storage -> the pack is refreshed in place (N1)
    prep_warm_ms = None
    if mode == "log_prob" and world == 1 and on_gpu:
        with torch.no_grad():
            for p_ in flow.parameters():
                p_.add_(0.0)                       # bumps the version counters: every prepared matrix is stale
        sync()
        t_w = time.perf_counter()
        eng.pack(dev)
        sync()
        prep_warm_ms = (time.perf_counter() - t_w) * 1e3
    # training modes: log_prob of head / middle / tail rows after the timed steps, device vs the CPU oracle at the CURRENT parameters
    train_parity = None
    if mode in ("train", "fit") and world == 1 and on_gpu and not args.no_cpu_baseline and D <= 1024:
        from oracle import usflows_oracle as orc      # the CPU oracle: parity leg only
        n3 = min(64, max(B // 3, 1))
        tidx = torch.cat([torch.arange(0, n3), torch.arange(B // 2, min(B // 2 + n3, B)), torch.arange(B - n3, B)]).unique()
        with torch.no_grad():
            got = flow.log_prob(x)[tidx.to(dev)].double().cpu()
            sd_now = {k: v.detach().cpu().clone() for k, v in flow.state_dict().items()}
            torch.set_num_threads(min(args.cpu_threads, os.cpu_count() or 1))
            ref_t = orc.flow_log_prob(sd_now, spec, x[tidx.to(dev)].cpu()).double()
        train_parity = {"rows": int(tidx.numel()), "placed_at": "head / middle / tail of the batch",
                        "max_rel_vs_cpu_fp32_oracle": float(((got - ref_t).abs() / ref_t.abs()).max().item()),
                        # (low-dimensional flows: a row's log_prob can sit near zero, where the row-wise relative figure says nothing)
                        "max_abs_err_over_max_abs_log_prob": float(((got - ref_t).abs().max() / ref_t.abs().max()).item()),
                        "what": "log_prob of the trained-on rows after the timed steps, device vs oracle/usflows_oracle.py at the current parameters"}
    probe_parity = None
    if probe is not None and lp is not None and mode == "log_prob":
        got = lp[probe[0]].double().cpu()
        probe_parity = {"rows": int(got.numel()), "placed_at": "head / middle / tail of the batch",
                        "max_rel_vs_reference_fp64": float(((got - probe[1].double()).abs() / probe[1].double().abs()).max().item()),
                        "reference": "golden rows of the REAL reference's fp64 run (tests/golden, committed fixture)"}
    headline = (args.config == "cfg2" and args.batch is None and args.dim is None and args.blocks is None
                and args.hidden is None and mode == "log_prob")
    metric = {"log_prob": "log_prob samples/sec (whole node), 32-layer 784-dim flow, batch 65536" if headline else
              f"log_prob samples/sec (whole node), {blocks}-layer {D}-dim flow, {args.config}",
              "sample": f"sample() samples/sec (whole node), {blocks}-layer {D}-dim flow",
              "train": f"training-step samples/sec (forward + backward + {'SophiaG' if args.optim == 'sophia' else 'Adam'} step), {blocks}-layer {D}-dim flow",
              "fit": f"Flow.fit optimiser-step samples/sec (forward + backward + {'SophiaG' if args.optim == 'sophia' else 'Adam'} step as Flow.fit issues it), {blocks}-layer {D}-dim flow"}[mode]
    if not on_gpu:
        metric = "[CPU PLUMBING TEST -- not a measurement] " + metric
    if world == 1 and args.force_dist:
        par = f"dp1 with the step's collective forced (one-rank {backend} group: the all-reduce is the identity, issued on the compute stream)"
    elif world == 1:
        par = "dp1 (single GPU: no collective)"
    elif mode == "sample":
        par = f"dp{world} (draws sharded by row range, disjoint Philox substreams, no collective)"
    elif mode in ("train", "fit"):
        par = f"dp{world} (batch sharded, one all-reduce of the flat gradient arena per step)"
    else:
        par = f"dp{world} (batch sharded, one all-reduce of 2 fp64 scalars [sum log_prob, count] per step)"
    out = {"metric": metric,
           "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
           "scaling": scaling, "vs_baseline": None,
           "dtype": ("f32" if (not on_gpu or eng.gemm_mode == "f32") else
                     "f32 (GEMMs as fp16x2 split on the f16 MFMA: 22 significant bits per operand, 3 products, fp32 accumulate)"
                     if eng.gemm_mode == "f16x2" else
                     "f32 (GEMMs as bf16x3 split on the bf16 MFMA: 24 significant bits, fp32 accumulate)"),
           "data": "synthetic",
           "world_size": world, "backend": backend, "device": str(dev),
           "config": {"workload": (f"{args.config} (experiments/synthetic/gaussian_mixture.yaml:40-93)" if args.config == "gm_live" else f"BASELINE {args.config}")
                                  + f": USFlow in_dims=[{D}], {blocks} additive coupling blocks, "
                                  f"{spec.conditioner}{hidden}+{'ReLU' if spec.negative_slope == 0 else f'LeakyReLU({spec.negative_slope})'}, "
                                  f"lu_transform=1, householder={args.householder}, affine_conjugation={args.conj}, "
                                  + (f"RadialDistribution(p=1, {'GammaMM x ' + str(spec.extra.get('gammamm_k')) if spec.radial_norm == 'gammamm' else 'LogNormal(6, .35)'}) base"
                                     if spec.base == "radial" else "Laplace(0,1) base")
                                  + (f", prior_scale={spec.extra['prior_scale']}" if spec.extra.get("prior_scale") is not None else "")
                                  + f"; {mode} of {B} rows per GPU "
                                  f"({global_rows} over {world} GPU(s)) resident in HBM; conditioned synthetic parameters "
                                  f"(seed {args.seed}, alpha 0.1)",
                      "rows_per_gpu": B, "global_rows": global_rows, "parallelism": par,
                      "fused_coupling": not args.unfused, "gemm_mode": eng.gemm_mode if on_gpu else "torch-cpu",
                      "merge_affine": (eng.merge_affine if eng.merge_affine != "auto" else
                                       f"auto -> {eng._merge_on('backward' if mode != 'sample' else 'forward')}; probe (accepted, element-wise, L1) "
                                       f"{eng.merge_guard_log[-1:]}") if on_gpu else None},
           "flow_algorithmic_tflops_per_gpu": round(flow_tflops, 2),
           "flow_frac_of_f32_mfma_peak": round(flow_tflops / F32_MFMA_PEAK_TFLOPS, 4),
           "param_prep_first_call_s": round(prep_s, 3),
           "param_prep_warm_ms": None if prep_warm_ms is None else round(prep_warm_ms, 2),
           "mean_log_prob": float(mean.item()) if mode != "sample" else None,
           "udl_check": udl, "probe_parity": probe_parity, "train_parity": train_parity,
           "roofline": roofline, "cpu_baseline": cpu, "fast_mode": fast}
    if mode == "fit":
        out["train_step"] = {"graph_replays": fit_replays[0], "of_steps": args.steps,
                             "what": "Flow.fit's optimiser step (zero grads, log_prob, backward, SophiaG update) on its own stream; "
                                     "captured as a hipGraph after three eager steps (batches <= 4096 rows)"}
    if under_launcher:
        dist.destroy_process_group()
    return out


# ----------------------------------------------------------------------------------------------------------------------
# image-shaped flows
# ----------------------------------------------------------------------------------------------------------------------
_IMAGE_KERNELS = ("usf_radial_logprob_f32", "usf_radial_logprob_grad_f32", "usf_conv2d_same_f32", "usf_conv2d_same_res_f32", "usf_channel_affine_f32", "usf_pointwise_conv_f32", "usf_layernorm_channels_f32",
                  "usf_gated_residual_f32", "usf_masked_residual_f32", "usf_base_logprob_f32", "usf_conv_wgrad_f32", "usf_conv2d_same_gate_f32",
                  "usf_layernorm_channels_bwd_f32", "usf_gated_residual_bwd_f32", "usf_gated_tail_f32", "usf_gated_tail_bwd_f32",
                  "usf_conv_wgrad_jobs_f32", "usf_partial_sum_jobs_f32", "usf_conv2d_weight_planes_batch_f32")
LAUNCH_BOUND_ROWS = 1024      # image training at or below this batch is a chain of dependent ~5 us launches: no kernel roofline applies
LAUNCH_FLOOR_US = 4.7         # a dependent launch inside the replayed graph (ordered kernel trace, profiles/r05_live_fit32_trace_after.txt)


def _image_launch_cost(name, a):
    """(class key, algorithmic flops, algorithmic HBM bytes) of one launch of an image-path entry point from its C
    arguments (usflows_amd/_ext.py wrappers).  Bytes: every operand tensor once; weights are KBs and not counted."""
    if name == "usf_conv2d_same_f32":        # (x, y, B, cin, cout, H, W, ks, planes, bias, in_mul, in_act, in_slope, out_act, out_slope, gate_x, gc, stream)
        B, cin, cout, H, W, ks, gc = int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), int(a[7]), int(a[16])
        flops = 2.0 * B * H * W * cin * cout * ks * ks
        byts = 4.0 * B * H * W * (cin + (gc if gc else cout) + (gc if gc else 0))        # gated: + the gate's skip input, C outputs
        return ("conv2d_same", cin, cout, ks, H, W, bool(gc)), flops, byts
    if name == "usf_conv2d_same_res_f32":    # (x, y, B, cin, cout, H, W, ks, planes, bias, in_mul, in_act, in_slope, res_x, res_mul, res_sign, stream)
        B, cin, cout, H, W, ks = int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), int(a[7])
        return ("conv2d_same+residual", cin, cout, ks, H, W), 2.0 * B * H * W * cin * cout * ks * ks, 4.0 * B * H * W * (cin + 2 * cout)
    if name == "usf_conv2d_same_gate_f32":   # (x, y, B, cin, cout, H, W, ks, planes, gate_h, gate_slope, gate_mul, stream)
        B, cin, cout, H, W, ks = int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), int(a[7])
        return ("conv2d_same+gate", cin, cout, ks, H, W), 2.0 * B * H * W * cin * cout * ks * ks, 4.0 * B * H * W * (cin + 2 * cout)
    if name == "usf_channel_affine_f32":     # (x, y, B, C, P, W, pre_sub, bias, stream)
        B, C, P = int(a[2]), int(a[3]), int(a[4])
        return ("channel_affine", C, P), 2.0 * B * P * C * C, 8.0 * B * P * C
    if name == "usf_pointwise_conv_f32":     # (x, y, B, cin, cout, P, W, bias, ..., gate_x, ...)
        B, cin, cout, P, gated = int(a[2]), int(a[3]), int(a[4]), int(a[5]), a[12] is not None
        cout_w = cout // 2 if gated else cout
        return ("pointwise_conv", cin, cout, P, gated), 2.0 * B * P * cin * cout, 4.0 * B * P * (cin + cout_w + (cout_w if gated else 0))
    if name == "usf_layernorm_channels_f32":
        B, C, P = int(a[2]), int(a[3]), int(a[4])
        return ("layernorm_channels", C, P), 8.0 * B * C * P, 8.0 * B * C * P
    if name == "usf_gated_residual_f32":     # (x, vg, y, B, CP, stream)
        B, CP = int(a[3]), int(a[4])
        return ("gated_residual", CP), 4.0 * B * CP, 16.0 * B * CP
    if name == "usf_masked_residual_f32":    # (x, t, mask, sign, y, B, CP, stream)
        B, CP = int(a[5]), int(a[6])
        return ("masked_residual", CP), 2.0 * B * CP, 12.0 * B * CP
    if name == "usf_base_logprob_f32":       # (z, ldz, M, D, ...)
        M, D = int(a[2]), int(a[3])
        return ("base_logprob", D), 4.0 * M * D, 4.0 * M * D
    if name == "usf_radial_logprob_f32":     # (z, ldz, M, D, ...): one read of the batch
        M, D = int(a[2]), int(a[3])
        return ("radial_logprob", D), 4.0 * M * D, 4.0 * M * D
    if name == "usf_radial_logprob_grad_f32":   # (z, ldz, r, g_lp, M, D, ...): read z, write d/dz, read it once more for d/dloc
        M, D = int(a[4]), int(a[5])
        return ("radial_logprob_grad", D), 6.0 * M * D, 12.0 * M * D
    if name == "usf_conv_wgrad_f32":         # (x, dy, B, cin, cout, H, W, ks, ...): exact fp32 on the f32 MFMA
        B, cin, cout, H, W, ks = int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), int(a[7])
        return ("conv_wgrad", cin, cout, ks, H, W), 2.0 * B * H * W * cin * cout * ks * ks, 4.0 * B * H * W * (cin + cout)
    if name == "usf_layernorm_channels_bwd_f32":   # (x, dy, dx, B, C, P, ...)
        B, C, P = int(a[3]), int(a[4]), int(a[5])
        return ("layernorm_channels_bwd", C, P), 16.0 * B * C * P, 12.0 * B * C * P
    if name == "usf_gated_residual_bwd_f32":       # (dy, vg, dvg, B, CP, stream)
        B, CP = int(a[3]), int(a[4])
        return ("gated_residual_bwd", CP), 8.0 * B * CP, 20.0 * B * CP
    if name == "usf_gated_tail_f32":               # (h, x, y, B, C, P, W, ...): 1 x 1 convolution C -> 2 C + gate + skip + layer norm
        B, C, P = int(a[3]), int(a[4]), int(a[5])
        return ("gated_tail", C, P), 2.0 * B * P * C * 2 * C, 12.0 * B * C * P
    if name == "usf_gated_tail_bwd_f32":           # (h, x, dy, dx, dh, dvg, B, C, P, ...): forward again + W^T product + dW
        B, C, P = int(a[6]), int(a[7]), int(a[8])
        return ("gated_tail_bwd", C, P), 3 * 2.0 * B * P * C * 2 * C, 20.0 * B * C * P
    return (name,), 0.0, 0.0


def main_image(args, under_launcher):
    """log_prob of an image-shaped flow (the reference's MNIST / CIFAR experiment models) over this rank's resident batch;
    the same contract as the flat configurations: W warm-up steps, K timed steps between barriers, max over ranks, ONE
    JSON line with the roofline of the dominant kernel and the CPU baseline (the image oracle, rank 0 at N = 1)."""
    import torch.distributed as dist
    cfg = IMAGE_CONFIGS[args.config]
    if args.device != "cuda":
        raise SystemExit("bench.py: the image configurations are GPU measurements (--device cuda)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1
    backend = None
    if under_launcher:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        backend = dist.get_backend()
    elif args.force_dist:               # one-rank group of its own: the collective path on ONE GPU (see main_flat)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        backend = dist.get_backend()
        under_launcher = True
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    from usflows_amd import _ext
    from usflows_amd.flows import USFlow
    from usflows_amd.networks import ConvNet2D
    from usflows_amd.parallel import mean_log_prob
    dims = list(cfg["in_dims"])
    base_kind = args.base or cfg.get("base", "laplace")
    radial_norm = args.radial_norm or cfg.get("radial_norm", "lognormal")
    prior_scale = args.prior_scale if args.prior_scale is not None else cfg.get("prior_scale")
    n_blocks = args.blocks or cfg["blocks"]

    def make_base(device):
        if base_kind == "laplace":
            return torch.distributions.Laplace(torch.zeros(dims, device=device), torch.ones(dims, device=device))
        from usflows_amd import distributions as UD
        # the YAMLs' constructor calls (mnist.yaml:79-92; fashionclasses_veriflow.yaml:79-93), on the CPU, then moved with the flow
        if radial_norm == "lognormal":
            nd = UD.LogNormal(loc=torch.ones([1]) * 6, scale=torch.ones([1]) * .35, device="cpu")
        else:
            nd = UD.GammaMM(concentration=torch.rand([20]) * 75, rate=torch.rand([20]), mixture_weights=torch.ones([20]) / 20, device="cpu")
        return UD.RadialDistribution(device="cpu", p=float("1"), loc=torch.zeros(dims), norm_distribution=nd)

    torch.manual_seed(args.seed)                                     # same parameters on every rank
    host = USFlow(make_base("cpu"), dims, n_blocks, ConvNet2D, dict(cfg["cond"]), prior_scale=prior_scale,
                  householder=cfg["householder"], affine_conjugation=True)
    _condition_image_flow(host, seed=args.seed)
    sd_cpu = {k: v.detach().clone() for k, v in host.state_dict().items()}
    torch.manual_seed(args.seed)
    flow = USFlow(make_base(dev if base_kind == "laplace" else "cpu"), dims, n_blocks,
                  ConvNet2D, dict(cfg["cond"]), prior_scale=prior_scale, householder=cfg["householder"], affine_conjugation=True)
    flow.load_state_dict(sd_cpu, strict=True)
    flow = flow.to(dev)
    del host
    if not _ext.lib_exists():
        raise RuntimeError("bench.py: libusflows_hip.so is missing -- refusing to time a fallback")
    B = args.batch or cfg["rows"]
    global_rows = B * world
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(B, *dims, generator=g).to(dev)                    # this rank's shard, resident in HBM
    acc = torch.zeros(2, dtype=torch.float64, device=dev)

    mode = args.mode or "log_prob"
    if mode not in ("log_prob", "train"):
        raise SystemExit("bench.py: the image configurations measure --mode log_prob (default) or train")
    opt = None
    if mode == "train":
        # one step of Flow.fit (flows.py:196-210) on the resident batch with the optimiser fit defaults to (SophiaG, flows.py:116)
        from usflows_amd.sophia import SophiaG
        opt = SophiaG(flow.parameters(), lr=1e-6) if args.optim == "sophia" else torch.optim.Adam(flow.parameters(), lr=1e-6)

    graph_steps = [0]

    def step():
        if mode == "train":
            # the optimiser step as Flow.fit issues it (flows.py:196-210): after three eager steps the step is captured as a
            # hipGraph and replayed (Flow._train_graph_step); the eager form serves until then
            # (data-parallel: eager steps -- the gradient all-reduce sits between backward and the optimiser's update)
            loss = flow._train_graph_step(opt, x, None) if not (args.eager_train or world > 1 or args.force_dist) else None
            if loss is not None:
                graph_steps[0] += 1
                return torch.tensor(-loss, dtype=torch.float64), None
            flow._zero_grad_for_step(opt)
            lp_ = flow.log_prob(x)
            loss = -lp_.mean() - flow.log_prior()              # Flow.fit's loss (flows.py:196-198)
            loss.backward()
            if world > 1 or args.force_dist:
                from usflows_amd.parallel import allreduce_gradients
                allreduce_gradients(flow, B)                   # ONE all-reduce of the flat gradient buffer, weighted by row counts
            opt.step()
            return -loss.detach().double(), lp_.detach()
        return mean_log_prob(flow, x, acc=acc, force_collective=args.force_dist)

    if mode == "train" and (world > 1 or args.force_dist):
        flow.__dict__["_grad_allreduce"] = (None, True)       # (Flow._zero_grad_for_step then zeroes the bound buffer in place)
    for _ in range(max(args.warmup, 5 if mode == "train" else 1)):
        mean, lp = step()
    if not args.no_kernel_timing and mode == "train" and not args.eager_train and world == 1 and not args.force_dist:
        # per-kernel times of the step: three eager steps with HIP events around every launch (a replayed graph runs no
        # host code to put events around), before the timed region
        _ext.launch_timing = {n: [] for n in _IMAGE_KERNELS}
        os.environ["USFLOWS_AMD_TRAIN_GRAPH"] = "0"
        keep = args.eager_train
        args.eager_train = True
        for _ in range(3):
            step()
        args.eager_train = keep
        os.environ.pop("USFLOWS_AMD_TRAIN_GRAPH", None)
        torch.cuda.synchronize()
        eager_timing, _ext.launch_timing = _ext.launch_timing, None
        eager_steps = 3
    else:
        eager_timing, eager_steps = None, 0
    graph_steps[0] = 0
    # per-launch HIP events on every TIMING_EVERY-th step of the timed region only (as for the flat configurations): an
    # instrumented step runs its ~230 calls one by one with two event records each instead of replaying the recorded op list
    # (round 5: the live MNIST configuration measured 41 ms per step with events on every step, its kernels add up to 32 ms)
    sampled = {n: [] for n in _IMAGE_KERNELS} if (not args.no_kernel_timing and eager_timing is None) else None
    n_sampled = 0
    if under_launcher:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i_step in range(args.steps):
        if sampled is not None:
            on = i_step % TIMING_EVERY == 0
            _ext.launch_timing = sampled if on else None
            n_sampled += 1 if on else 0
        mean, lp = step()
    torch.cuda.synchronize()
    if under_launcher:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    _ext.launch_timing = None
    timing = sampled
    n_timed = max(n_sampled, 1)
    if eager_timing is not None:
        timing, n_timed = eager_timing, eager_steps
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if under_launcher:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank != 0:
        if under_launcher:
            dist.destroy_process_group()
        return None
    ms_per_step = elapsed / args.steps * 1e3
    value = global_rows * args.steps / elapsed

    roofline = None
    if timing and any(timing.values()):
        classes = {}
        for name, recs in timing.items():
            for e0, e1, a in recs:
                key, fl, by = _image_launch_cost(name, a)
                c = classes.setdefault(key, dict(ms=0.0, n=0, flops=fl, bytes=by))
                c["ms"] += e0.elapsed_time(e1)
                c["n"] += 1
        dom = max(classes, key=lambda k: classes[k]["ms"])
        c = classes[dom]
        avg_ms = c["ms"] / c["n"]
        mfma_peak = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1)
        if dom[0] == "conv_wgrad":
            mfma_peak = F32_MFMA_PEAK_TFLOPS                    # exact fp32 on v_mfma_f32_16x16x4_f32
        t_mfma = c["flops"] / (mfma_peak * 1e12) if dom[0].startswith("conv") else 0.0   # only the convolutions run on the matrix cores
        t_hbm = c["bytes"] / (HBM_PEAK_GBS * 1e9)
        total_ms = sum(v["ms"] for v in classes.values())
        per_kernel = {"/".join(str(p_) for p_ in k): {"ms_per_step": round(v["ms"] / n_timed, 3), "launches_per_step": v["n"] // n_timed,
                                                      "hbm_frac": round(v["bytes"] / (v["ms"] / v["n"] * 1e-3) / (HBM_PEAK_GBS * 1e9), 3)}
                      for k, v in sorted(classes.items(), key=lambda kv: -kv[1]["ms"])}
        if t_mfma >= t_hbm:
            ach = c["flops"] / (avg_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(ach / mfma_peak, 4),
                        "peak_is": "f32-input MFMA peak (v_mfma_f32_16x16x4_f32, exact fp32)" if dom[0] == "conv_wgrad" else
                                   "dense bf16 MFMA peak (2500) / 6 products per fp32 product (bf16x3 split)"}
        else:
            ach = c["bytes"] / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "peak_is": "HBM3E ~8 TB/s"}
        roofline.update({"traffic": None, "kernel": "/".join(str(p_) for p_ in dom),
                         "both_roofs": {"mfma_time_ms": round(t_mfma * 1e3, 4), "hbm_time_ms": round(t_hbm * 1e3, 4)},
                         "measured_by": f"HIP events around every launch of every {TIMING_EVERY}th step of this run's timed region",
                         "avg_launch_ms": round(avg_ms, 4), "launches": c["n"], "share_of_gpu_time": round(c["ms"] / total_ms, 3),
                         "algorithmic_flops_per_launch": c["flops"], "algorithmic_bytes_per_launch": c["bytes"],
                         "all_kernels": per_kernel, "kernel_ms_per_step": round(total_ms / n_timed, 3)})
        if eager_timing is not None:
            roofline["measured_by"] = ("HIP events around every launch of three eager steps of this run, before the timed region "
                                       "(the timed steps are hipGraph replays, as in Flow.fit)")
        if mode == "train" and B <= LAUNCH_BOUND_ROWS:
            # a kernel roofline fraction says nothing about a step that is a chain of dependent launches of a
            # few microseconds each -- what bounds it is their NUMBER times the launch floor
            n_usf = sum(v["n"] for v in classes.values()) // n_timed
            roofline.update({"bound": "launch", "frac": None, "achieved": None, "peak": None, "unit": None,
                             "peak_is": "no kernel roofline applies: the step is bound by the number of dependent launches",
                             "launch_bound": {"library_launches_per_step": n_usf, "floor_us_per_launch": LAUNCH_FLOOR_US,
                                              "floor_ms": round(n_usf * LAUNCH_FLOOR_US * 1e-3, 3), "ms_per_step": round(ms_per_step, 3),
                                              "note": "torch's own launches of the loss / log-det / optimiser glue (~50 per step) are not "
                                                      "counted; ordered trace of one replayed step: profiles/r05_live_fit32_trace_after.txt"}})

    # ---- CPU baseline + parity: the image oracle (torch-CPU restatement of the reference's image path) on a bounded
    # sample of the same rows; rank 0 at N = 1 only ----
    cpu = None
    if world == 1 and not args.no_cpu_baseline and mode == "train":
        with torch.no_grad():               # (outside the timed region) the trained-on rows' log_prob for the parity check
            lp = flow.log_prob(x)
    if world == 1 and not args.no_cpu_baseline and lp is not None:
        from oracle import usflows_image_oracle as iorc           # the CPU oracle: this leg only
        spec = iorc.ImageSpec(in_dims=dims, coupling_blocks=n_blocks, cond_args=dict(cfg["cond"]), householder=cfg["householder"],
                              affine_conjugation=True, base=base_kind, radial_p=1.0, radial_norm=radial_norm)
        torch.set_num_threads(min(args.cpu_threads, os.cpu_count() or 1))
        rows = min(B, args.cpu_rows * 2)
        # head, middle and tail of the batch
        idx = torch.cat([torch.arange(0, rows // 2), torch.arange(B // 2, B // 2 + rows // 4), torch.arange(B - rows // 4, B)]).unique()
        if mode == "train":
            # (a training step's parameters have moved by lr = 1e-6 per step since sd_cpu was taken: the oracle sees the current ones)
            sd_cpu = {k: v.detach().cpu().clone() for k, v in flow.state_dict().items()}
        xc = x[idx.to(dev)].cpu()
        with torch.no_grad():
            iorc.flow_log_prob(sd_cpu, spec, xc[:32])
            n_it, t_c = 0, time.perf_counter()
            while True:
                ref = iorc.flow_log_prob(sd_cpu, spec, xc)
                n_it += 1
                if time.perf_counter() - t_c > args.cpu_seconds or n_it >= 20:
                    break
            cpu_s = time.perf_counter() - t_c
        rel = ((lp[idx.to(dev)].cpu().double() - ref.double()).abs() / ref.double().abs())
        cpu = {"value": round(xc.shape[0] * n_it / cpu_s, 1), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{n_it} x log_prob of {xc.shape[0]} rows of the batch (head, middle, tail) through oracle/usflows_image_oracle.py, {cpu_s:.1f} s",
               "host_cpus": os.cpu_count(), "parity_rows": int(rel.numel()), "parity_max_rel_vs_cpu_fp32": float(rel.max().item())}
        if mode == "train":
            cpu["note"] = ("the CPU oracle restates log_prob only: value = the oracle's log_prob rate; parity = log_prob of the "
                           "trained-on rows after the timed steps, device vs oracle with the current parameters")

    hw = dims[1] * dims[2]
    cnd = cfg["cond"]
    what = "log_prob" if mode == "log_prob" else "training-step (log_prob + backward + optimiser step)"
    out = {"metric": f"{what} samples/sec (whole node), image flow {args.config}" + (f" ({base_kind} base)" if base_kind != "laplace" else ""), "value": round(value, 1), "unit": "samples/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 (3 x 3 convolutions as bf16x3 split on the bf16 MFMA: 24 significant bits, fp32 accumulate; everything else fp32 VALU)",
           "data": "synthetic", "world_size": world, "backend": backend, "device": str(dev),
           "config": {"workload": f"{args.config}: USFlow in_dims={dims} ({cfg['ref']}), {n_blocks} additive coupling blocks, "
                                  f"ConvNet2D(c_hidden {cnd['c_hidden']}, {cnd['num_layers']} gated layer(s), layer norm, ReLU), lu_transform=1, "
                                  f"householder={cfg['householder']}, affine_conjugation=True, prior_scale={prior_scale}, "
                                  + ("Laplace(0,1) base (the first goldens' base; the live YAMLs draw a radial base: --base radial)"
                                     if base_kind == "laplace" else
                                     f"RadialDistribution(zeros{dims}, p=1, {'LogNormal(6, .35)' if radial_norm == 'lognormal' else 'GammaMM x 20'}) base "
                                     f"with trainable loc and norm parameters (the live YAMLs' base), density and gradients on usf_radial_logprob(_grad)_f32")
                                  + f"; {what} of {B} rows per GPU ({global_rows} over {world} GPU(s)) "
                                  f"resident in HBM; conditioned synthetic parameters (seed {args.seed})",
                      "rows_per_gpu": B, "global_rows": global_rows, "pixels": hw,
                      "parallelism": "dp1 (single GPU: no collective)" if world == 1 else
                                     f"dp{world} (batch sharded, one all-reduce of 2 fp64 scalars [sum log_prob, count] per step)"},
           "mean_log_prob": float(mean.item()), "roofline": roofline, "cpu_baseline": cpu}
    if mode == "train":
        out["train_step"] = {"graph_replays": graph_steps[0], "of_steps": args.steps,
                             "what": "Flow.fit's optimiser step (zero grads, log_prob, backward, SophiaG update) on the resident batch"}
    if under_launcher:
        dist.destroy_process_group()
    return out


def _condition_image_flow(flow, seed, alpha=0.3):
    """well-conditioned parameters for an image flow built with default initialisation (the default
    initialisation explodes): L <- I + alpha tril(L, -1); U <- alpha triu(U, 1) + diag(+-[0.75, 1.25]); scale <- +-[0.5, 1.5]"""
    from usflows_amd import transforms as T
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in flow.modules():
            if isinstance(m, T.LUTransform):
                D = m.L_raw.shape[0]
                m.L_raw.copy_(torch.eye(D) + alpha * m.L_raw.tril(-1))
                sign = torch.where(torch.rand(D, generator=g) < 0.5, -1.0, 1.0)
                m.U_raw.copy_(alpha * m.U_raw.triu(1) + torch.diag(sign * (0.75 + 0.5 * torch.rand(D, generator=g))))
            if isinstance(m, T.ScaleTransform):
                sign = torch.where(torch.rand(m.scale.shape, generator=g) < 0.5, -1.0, 1.0)
                m.scale.copy_(sign * (0.5 + torch.rand(m.scale.shape, generator=g)))


def _variant_code(M, N, K, split):
    """which usf_linear_f32 instantiation served this shape (usf_linear_variant: dispatch logic only, no launch)"""
    import ctypes as C
    from usflows_amd import _ext
    d = _ext.LinearDesc()
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc = M, N, K, K, K, N
    d.A = d.W = d.C = 1 << 20
    if split:
        d.W_split, d.ldw_split = 1 << 20, (K + 31) // 32 * 32
        d.split_plane_stride = N * d.ldw_split
    return _ext.load().usf_linear_variant(C.byref(d))


def _variant_name(code):
    if code // 1000 == 3:
        c = code % 1000
        return f"linear_bf16x3_kernel<{c // 100},{c // 10 % 10},{c % 10}>"
    if code // 1000 == 2:
        c = code % 1000
        return f"linear_kernel<{c // 100},{c // 10 % 10},{c % 10},16>"
    return "linear_skinny_kernel"


def _from_profile(kind, base, field, args, B, D, hidden):
    """(value, source) from the newest committed profiles/rNN_<kind>.json covering this kernel at the cfg2 shape"""
    if not (B == 65536 and D == 784 and list(hidden) == [256, 256] and args.config == "cfg2"):
        return None, None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{kind}.json")), reverse=True):
        try:
            prof = json.load(open(path))
            cand = [(v.get("dispatches", 0), v) for k, v in prof["kernels"].items() if k.split("<")[0] == base]
            if cand:
                rel = os.path.relpath(path, ROOT)
                return max(cand, key=lambda c: c[0])[1][field], (
                    f"{rel}: separate rocprofv3 --pmc pass of this command on another run/box "
                    f"(NOT measured by this process)")
        except Exception:
            continue
    return None, None


if __name__ == "__main__":
    main()
