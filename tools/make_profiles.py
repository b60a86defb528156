"""Regenerate the committed profile summaries (run on the GPU box from the repo root):

    python3 tools/make_profiles.py r03 [sections]   # writes gpurun_out/profiles_r03/*, copy them into profiles/
    sections (default all): bench pmc f16 peak other train trainpmc image imagetrain live -- several calls fit gpurun's 20-minute limit

1. `python3 bench.py` (defaults)                                   -> <tag>_bench.json
2. `rocprofv3 --kernel-trace --stats -- python3 bench.py`          -> <tag>_bench_kernel_stats.{csv,md}, <tag>_bench_under_rocprof.json
3. `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE` on a short bench run
                                                                   -> <tag>_hbm_traffic.json
3b. `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... GRBM_GUI_ACTIVE` (own pass)        -> <tag>_mfma_util.json
4. `bench.py --mode train` at 128 / 4096 / 65536 rows (+ the composite path at 4096), kernel stats of the training run
                                                                   -> <tag>_train_bench.jsonl, <tag>_train_kernel_stats.md
5. image-shaped flows: `bench.py --config mnist_image / cifar_image`, kernel trace and PMC passes of the MNIST model
                                                                   -> <tag>_bench_{mnist,cifar}_image.json, <tag>_image_*.{md,json}
6. image-shaped flows, training: `bench.py --config mnist_image / cifar_image --mode train` at several batch sizes (+ the torch
   autograd path), kernel trace and PMC passes of eager steps     -> <tag>_image_train_bench.jsonl, <tag>_image_train_*.{md,json}
This script never touches the GPU itself (every step is a child process), so the profiler wraps the
program directly.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}")
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")


def last_json_line(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return line
    raise RuntimeError("no JSON line in bench output:\n" + text[-2000:])


def run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, **kw)



SECTIONS = sys.argv[2:] or ["bench", "pmc", "f16", "peak", "other", "train", "trainpmc", "image", "imagetrain", "live"]


B, D, H = 65536, 784, 256          # cfg2 shape (algorithmic byte counts of the traffic summaries)


def want(name):
    return name in SECTIONS


# 3b. matrix-pipe utilisation (own PMC pass: SQ + GRBM counters, no tracing options)
def mfma_util(csv_path, out_path):
    agg2 = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(csv_path)):
        k = row["Kernel_Name"]
        if "usf::" not in k:
            continue
        k = k[k.index("usf::") + 5:]
        k = k[: k.index("(")] if "(" in k else k
        agg2[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k, dct in sorted(agg2.items()):
        m = {c: sum(v) / len(v) for c, v in dct.items()}
        if not m.get("GRBM_GUI_ACTIVE"):
            continue
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0                      # the counter sums the 8 XCDs
        res[k] = {"dispatches": len(dct["GRBM_GUI_ACTIVE"]), "gpu_cycles": round(cyc),
                  "mfma_busy_cycles": m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
                  "mfma_util": round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0), 4),
                  "wave_quad_cycles": {c: m.get(c) for c in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY",
                                                             "SQ_WAIT_ANY")}}
    json.dump({
        "source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                  "SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE on `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                  "--no-kernel-timing` (own pass, no tracing options), MI355X; tools/make_profiles.py",
        "units": "means per dispatch.  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (gpu_cycles x 1024 SIMDs): fraction of all matrix-pipe "
                 "cycles spent in an MFMA at the clock the kernel actually ran at (gpu_cycles = GRBM_GUI_ACTIVE / 8 XCDs; "
                 "SQ_VALU_MFMA_BUSY_CYCLES = 16 per v_mfma_f32_16x16x32_bf16, exactly the kernel's MFMA count x 16).  "
                 "Times achieved clock / 2.4 GHz it is the fraction of the 2.5 PFLOP/s bf16 peak, i.e. bench.py's roofline.frac.  "
                 "wave_quad_cycles: SQ_WAIT_INST_ANY = issue stalls (matrix-pipe dependency / pipe busy), SQ_WAIT_ANY = parked at "
                 "s_waitcnt / barriers, SQ_ACTIVE_INST_ANY = issuing; they sum to ~SQ_WAVE_CYCLES (MI355X_MICROARCH.md).",
        "kernels": res}, open(out_path, "w"), indent=1)


if want("bench"):
    # 1. plain bench (the default line: headline + the "also" block)
    r = run(["python3", "bench.py"])
    open(os.path.join(out, f"{tag}_bench.json"), "w").write(last_json_line(r.stdout) + "\n")

    # 2. kernel trace + stats
    d = os.path.join(out, "ktrace")
    r = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--no-also"])
    open(os.path.join(out, f"{tag}_bench_under_rocprof.json"), "w").write(last_json_line(r.stdout) + "\n")
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, f"{tag}_bench_kernel_stats.csv"), "w") as f:
        f.write(open(stats[0]).read())
    with open(os.path.join(out, f"{tag}_bench_kernel_stats.md"), "w") as f:
        f.write(f"# {tag}: rocprofv3 --kernel-trace --stats of `python3 bench.py --no-also`\n\n"
                "Command (GPU box): `rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 bench.py --no-also`\n"
                "(--no-also: the headline workload alone, without the other configurations the default line appends)\n"
                "(defaults: --gpus 1 --steps 10 --warmup 3, gemm_mode bf16x3, planes pipeline; per pass 1 pack + 33 affine GEMMs on\n"
                "planes (the last with fp32 output) + 32 fused couplings on planes + 1 tail; the same process then measures the\n"
                "opt-in fp16x2 mode (the <2, ...> instantiations); the CPU-baseline leg and the first (parameter-prep) call add\n"
                "the torch/rocBLAS kernels at the bottom)\n\n"
                "| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|\n")
        for row in rows[:14]:
            name = row["Name"]
            name = name if len(name) < 110 else name[:107] + "..."
            f.write(f"| `{name}` | {row['Calls']} | {row['TotalDurationNs']} | {float(row['AverageNs']):.0f} | {row['Percentage']} |\n")
        f.write(f"\nbench.py's own HIP-event measurement of the same kind of run: {tag}_bench.json (un-profiled) and\n"
                f"{tag}_bench_under_rocprof.json (this run); `roofline.avg_launch_ms` there is the per-launch average of the\n"
                "dominant kernel above.\n")

if want("pmc"):
    # 3. PMC passes (each counter in its own run; no tracing options beside --pmc)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, "pmc_" + ctr)
        run(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--steps", "2",
             "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing", "--no-fast-mode"])
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                k = row["Kernel_Name"]
                if "usf::" not in k:
                    continue
                k = k[k.index("usf::") + 5:]
                k = k[: k.index("(")] if "(" in k else k
                agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))

    alg = {"gemm_planes_kernel": 2 * B * 800 * 6 + 3 * 800 * 800 * 2,          # planes in + planes out + weight planes (bf16x3)
           "coupling_planes_kernel": 3 * 13 * 32 * B * 6 + 3 * 2 * (H * 416 + H * H + 416 * H),
           "linear_bf16x3_kernel": 2 * B * D * 4 + 3 * D * 800 * 2,
           "coupling_bf16x3_kernel": (B * D + B * (D // 2)) * 4 + 3 * 2 * (H * 416 + H * H + 416 * H),
           "base_logprob_kernel": B * D * 4 + B * 4,
           "pack_planes_rows_kernel": B * D * 4 + B * 800 * 6}                # fp32 rows in, bf16x3 planes out
    kern = {}
    for k, dct in sorted(agg.items()):
        f = sum(dct["FETCH_SIZE"]) / max(1, len(dct["FETCH_SIZE"]))
        w = sum(dct["WRITE_SIZE"]) / max(1, len(dct["WRITE_SIZE"]))
        base = k.split("<")[0]
        kern[k] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "dispatches": len(dct["FETCH_SIZE"]),
                   "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                   "algorithmic_bytes_per_launch": alg.get(base)}
    json.dump({
        "source": "rocprofv3 --pmc FETCH_SIZE and (separate pass) --pmc WRITE_SIZE on `python3 bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-kernel-timing` (default gemm_mode bf16x3), MI355X; tools/make_profiles.py",
        "units": "counter values are KB per dispatch (mean over dispatches). hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: "
                 "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B); calibration inside "
                 "the same run: pack_planes_rows_kernel<3, false> reads its 205.5 MB of fp32 rows once, coalesced (its FETCH_SIZE should read "
                 "0.50x of that), and writes 314.6 MB of planes (its WRITE_SIZE should read 1.00x of that). "
                 "algorithmic_bytes_per_launch: activations in + out once, weights once (the last GEMM, <3, 5, true>, stores [B, 8] "
                 "partial sums of the base density instead of rows since round 5).",
        "kernels": kern}, open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w"), indent=1)

    d = os.path.join(out, "pmc_mfma")
    run(["rocprofv3", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
         "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--steps", "2",
         "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing", "--no-fast-mode"])
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        mfma_util(fn, os.path.join(out, f"{tag}_mfma_util.json"))

if want("f16"):
    # 3c. the opt-in fp16x2 mode (planes pipeline): bench line, kernel stats, matrix-pipe utilisation, HBM traffic
    r = run(["python3", "bench.py", "--gemm", "f16x2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"])
    open(os.path.join(out, f"{tag}_bench_f16x2.json"), "w").write(last_json_line(r.stdout) + "\n")
    d = os.path.join(out, "ktrace_f16x2")
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--gemm", "f16x2",
         "--no-cpu-baseline"])
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, f"{tag}_bench_f16x2_kernel_stats.md"), "w") as f:
        f.write(f"# {tag}: rocprofv3 --kernel-trace --stats of `python3 bench.py --gemm f16x2 --no-cpu-baseline`\n\n"
                "(opt-in fast mode: planes pipeline with fp16x2 planes; per pass 1 pack + 32 affine GEMMs on planes + 32 x 3\n"
                "conditioner GEMMs on planes + the last affine with fp32 output + 1 tail)\n\n"
                "| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|\n")
        for row in rows[:10]:
            name = row["Name"]
            name = name if len(name) < 110 else name[:107] + "..."
            f.write(f"| `{name}` | {row['Calls']} | {row['TotalDurationNs']} | {float(row['AverageNs']):.0f} | {row['Percentage']} |\n")
    d = os.path.join(out, "pmc_mfma_f16x2")
    run(["rocprofv3", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
         "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--gemm", "f16x2",
         "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing"])
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        mfma_util(fn, os.path.join(out, f"{tag}_mfma_util_f16x2.json"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, "pmc_f16x2_" + ctr)
        run(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--gemm", "f16x2", "--steps",
             "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing"])
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                k = row["Kernel_Name"]
                if "usf::" not in k:
                    continue
                k = k[k.index("usf::") + 5:]
                k = k[: k.index("(")] if "(" in k else k
                agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kern = {}
    for k, dct in sorted(agg.items()):
        f_ = sum(dct["FETCH_SIZE"]) / max(1, len(dct["FETCH_SIZE"]))
        w_ = sum(dct["WRITE_SIZE"]) / max(1, len(dct["WRITE_SIZE"]))
        kern[k] = {"FETCH_SIZE_KB": round(f_, 1), "WRITE_SIZE_KB": round(w_, 1), "dispatches": len(dct["FETCH_SIZE"]),
                   "hbm_bytes_per_launch": int((2 * f_ + w_) * 1024),
                   "algorithmic_bytes_per_launch": {"gemm_planes_kernel": 2 * B * 800 * 4 + 2 * 800 * 800 * 2,
                                                    "coupling_planes_kernel": 3 * 13 * 32 * B * 4 + 2 * 2 * (H * 416 + H * H + 416 * H)
                                                    }.get(k.split("<")[0])}
    json.dump({
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python3 bench.py --gemm f16x2 --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-kernel-timing`, MI355X; tools/make_profiles.py",
        "units": "as in the bf16x3 file (fp16x2 planes: 4 bytes per element)",
        "kernels": kern}, open(os.path.join(out, f"{tag}_hbm_traffic_f16x2.json"), "w"), indent=1)

if want("peak"):
    # 3d. what the chip sustains on the split-precision instruction mix (register / LDS operand loops; tools/exp_mfma_peak.hip)
    exe = os.path.join(ROOT, "tools", "exp_mfma_peak")
    if os.path.exists(exe):
        r = run([exe, "400"])
        json.dump({"source": "tools/exp_mfma_peak 400 (hipcc -O3 --offload-arch=gfx950 tools/exp_mfma_peak.hip), MI355X: register-only "
                             "loops of the bf16x3 instruction mix at the occupancy of the GEMM kernels; TF = bf16 MFMA flops, "
                             "fp32-equiv = / 6; the clock column is s_memtime / s_memrealtime x 100 MHz and reads ~14 % high on "
                             "this part (the real-time counter ticks slower than 100 MHz: 2054 TF at 100 % pipe occupancy is "
                             "1.97 GHz)",
                   "lines": r.stdout.strip().splitlines()}, open(os.path.join(out, f"{tag}_mfma_peak.json"), "w"), indent=1)

if want("other"):
    # 3e. the other BASELINE configurations on one GPU
    lines = []
    for cfg, extra in (("cfg3", ["--steps", "5", "--warmup", "2"]), ("cfg4", ["--steps", "3", "--warmup", "1"]),
                       ("cfg5", ["--steps", "3", "--warmup", "1"])):
        r = run(["python3", "bench.py", "--config", cfg, "--no-cpu-baseline"] + extra)
        lines.append(last_json_line(r.stdout))
    open(os.path.join(out, f"{tag}_bench_other_configs.jsonl"), "w").write("\n".join(lines) + "\n")

if want("train"):
    # 4. training step and parameter prep (N1): bench lines + kernel stats of the training run
    lines = []
    for extra in (["--mode", "train", "--batch", "128"], ["--mode", "train", "--batch", "4096"], ["--mode", "train"]):
        r = run(["python3", "bench.py", "--steps", "10", "--warmup", "3"] + extra)
        lines.append(last_json_line(r.stdout))
        for composite_env in ([("USFLOWS_AMD_TRAIN", "composite")] if extra[-1] == "4096" else []):
            r = subprocess.run(["python3", "bench.py", "--steps", "5", "--warmup", "2"] + extra, cwd=ROOT,
                               env=dict(env, **dict([composite_env])), capture_output=True, text=True)
            j = json.loads(last_json_line(r.stdout))
            j["config"]["training_path"] = "composite torch ops (USFLOWS_AMD_TRAIN=composite)"
            lines.append(json.dumps(j))
    open(os.path.join(out, f"{tag}_train_bench.jsonl"), "w").write("\n".join(lines) + "\n")
    # Flow.fit itself at the reference's batch sizes (the step replayed as a hipGraph): tools/fit_small_batch.py
    with open(os.path.join(out, f"{tag}_train_small_batch.txt"), "w") as f:
        for bsz, steps in (("32", "60"), ("256", "30")):
            r = run(["python3", "tools/fit_small_batch.py", bsz, steps])
            f.write("\n".join(l for l in r.stdout.splitlines() if l.startswith("Flow.fit")) + "\n")
    d = os.path.join(out, "ktrace_train")
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--mode",
         "train", "--steps", "5", "--warmup", "2"])
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, f"{tag}_train_kernel_stats.md"), "w") as f:
        f.write(f"# {tag}: rocprofv3 --kernel-trace --stats of `python3 bench.py --mode train --steps 5 --warmup 2`\n\n"
                "One step = Flow.log_prob under autograd (device forward with saved activations) + hand-derived device backward\n"
                "+ SophiaG step (the reference's Flow.fit default), cfg2 model, 65536 rows; 7 steps in the trace (+ the first,\n"
                "recording, one).\n\n"
                "| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|\n")
        for row in rows[:18]:
            name = row["Name"]
            name = name if len(name) < 110 else name[:107] + "..."
            f.write(f"| `{name}` | {row['Calls']} | {row['TotalDurationNs']} | {float(row['AverageNs']):.0f} | {row['Percentage']} |\n")

if want("trainpmc"):
    # 4b. HBM traffic of the training step's kernels (round 5: the planes training path), each counter in its own pass
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, "pmc_train_" + ctr)
        run(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--mode", "train", "--steps", "2",
             "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing"])
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                k = row["Kernel_Name"]
                if "usf::" not in k:
                    continue
                k = k[k.index("usf::") + 5:]
                k = k[: k.index("(")] if "(" in k else k
                # (the weight-gradient kernel serves four shapes: keyed by its grid as well)
                agg[k + " grid " + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kern = {}
    for k, dct in sorted(agg.items()):
        f = sum(dct["FETCH_SIZE"]) / max(1, len(dct["FETCH_SIZE"]))
        w = sum(dct["WRITE_SIZE"]) / max(1, len(dct["WRITE_SIZE"]))
        kern[k] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "dispatches": len(dct["FETCH_SIZE"]),
                   "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump({
        "source": "rocprofv3 --pmc FETCH_SIZE and (separate pass) --pmc WRITE_SIZE on `python3 bench.py --mode train --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-kernel-timing` (cfg2 model, 65536 rows, planes training path), MI355X; tools/make_profiles.py",
        "units": "KB per dispatch (mean); hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE doubled as "
                 "MI355X_MICROARCH.md prescribes for gfx950)",
        "algorithmic_bytes": {"wgrad 784 x 784 (usf_wgrad_blocked_f32)": 2 * B * 800 * 6, "gemm_planes 784 x 784": 2 * B * 800 * 6 + 3 * 800 * 800 * 2,
                              "conditioner wgrads of one coupling": (392 + 256 + 256 + 256 + 256 + 400) * B * 6},
        "kernels": kern}, open(os.path.join(out, f"{tag}_train_hbm_traffic.json"), "w"), indent=1)

# 5. image-shaped flows: bench lines of the reference's MNIST / CIFAR experiment models, kernel trace and the
#    PMC passes (FETCH_SIZE / WRITE_SIZE and the matrix-pipe counters, each in its own run) of the MNIST configuration
if want("image"):
    for cfg, steps in (("mnist_image", "10"), ("cifar_image", "5")):
        r = run(["python3", "bench.py", "--config", cfg, "--steps", steps])
        open(os.path.join(out, f"{tag}_bench_{cfg}.json"), "w").write(last_json_line(r.stdout) + "\n")
    d = os.path.join(out, "ktrace_image")
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--config",
         "mnist_image", "--steps", "10", "--no-cpu-baseline"])
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, f"{tag}_image_kernel_stats.md"), "w") as f:
        f.write(f"# {tag}: rocprofv3 --kernel-trace --stats of `python3 bench.py --config mnist_image --steps 10 --no-cpu-baseline`\n\n"
                "(log_prob of 65 536 rows of the reference's MNIST experiment model, tests/explib/mnist.yaml:44-77: 13 calls in the\n"
                "trace -- 3 warm-up + 10 timed; per call 6 convolutions on the matrix cores (the last one of each coupling with the\n"
                "residual in its output stream), 2 gated pointwise passes, 3 channel affine launches (composed runs), 1 scale pass,\n"
                "1 base density)\n\n"
                "| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|\n")
        for row in rows[:12]:
            name = row["Name"]
            name = name if len(name) < 110 else name[:107] + "..."
            f.write(f"| `{name}` | {row['Calls']} | {row['TotalDurationNs']} | {float(row['AverageNs']):.0f} | {row['Percentage']} |\n")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, "pmc_image_" + ctr)
        run(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--config", "mnist_image",
             "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing"])
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                k = row["Kernel_Name"]
                if "usf::" not in k:
                    continue
                k = k[k.index("usf::") + 5:]
                k = k[: k.index("(")] if "(" in k else k
                agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kern = {}
    for k, dct in sorted(agg.items()):
        f_ = sum(dct["FETCH_SIZE"]) / max(1, len(dct["FETCH_SIZE"]))
        w_ = sum(dct["WRITE_SIZE"]) / max(1, len(dct["WRITE_SIZE"]))
        kern[k] = {"FETCH_SIZE_KB": round(f_, 1), "WRITE_SIZE_KB": round(w_, 1), "dispatches": len(dct["FETCH_SIZE"]),
                   "hbm_bytes_per_launch": int((2 * f_ + w_) * 1024)}
    json.dump({
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python3 bench.py --config mnist_image --steps 2 "
                  "--warmup 1 --no-cpu-baseline --no-kernel-timing`, MI355X; tools/make_profiles.py",
        "units": "means per dispatch over ALL shapes a kernel served (the convolution kernel: three layer shapes); "
                 "hbm_bytes_per_launch = (2 FETCH_SIZE + WRITE_SIZE) KB: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide "
                 "coalesced reads on gfx950.  Algorithmic bytes of the three convolutions at 65 536 rows of 7 x 7 pixels: 16 -> 32: "
                 "616 MB, 32 -> 32: 822 MB, 32 -> 16: 616 MB (input + output once, fp32).",
        "kernels": kern}, open(os.path.join(out, f"{tag}_image_hbm_traffic.json"), "w"), indent=1)
    d = os.path.join(out, "pmc_image_mfma")
    run(["rocprofv3", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
         "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--config",
         "mnist_image", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing"])
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        mfma_util(fn, os.path.join(out, f"{tag}_image_mfma_util.json"))

# 6. image-shaped flows, training step (SURVEY rows N2 x N4): bench lines, kernel trace of eager steps, PMC passes
if want("imagetrain"):
    with open(os.path.join(out, f"{tag}_image_train_bench.jsonl"), "w") as f:
        for cfg, batch in (("mnist_image", "65536"), ("mnist_image", "4096"), ("mnist_image", "32"), ("cifar_image", "16384"),
                           ("cifar_image", "32")):
            r = run(["python3", "bench.py", "--config", cfg, "--mode", "train", "--batch", batch, "--steps", "10"])
            f.write(last_json_line(r.stdout) + "\n")
        # the same flow through torch autograd + MIOpen (what round 2 replayed): USFLOWS_AMD_TUNE=image_train=0
        env_t = dict(env, USFLOWS_AMD_TUNE="image_train=0")
        for cfg, batch in (("mnist_image", "65536"), ("mnist_image", "32")):
            print("+ USFLOWS_AMD_TUNE=image_train=0 python3 bench.py --config", cfg, "--mode train --batch", batch, flush=True)
            r = subprocess.run(["python3", "bench.py", "--config", cfg, "--mode", "train", "--batch", batch, "--steps", "10"], cwd=ROOT,
                               env=env_t, capture_output=True, text=True)
            line = json.loads(last_json_line(r.stdout))
            line["config"]["workload"] += " [USFLOWS_AMD_TUNE=image_train=0: torch autograd + MIOpen backward]"
            f.write(json.dumps(line) + "\n")
    d = os.path.join(out, "ktrace_image_train")
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--config",
         "mnist_image", "--mode", "train", "--eager-train", "--steps", "10", "--no-kernel-timing"])
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, f"{tag}_image_train_kernel_stats.md"), "w") as f:
        f.write(f"# {tag}: rocprofv3 --kernel-trace --stats of `python3 bench.py --config mnist_image --mode train --eager-train --steps 10 --no-kernel-timing`\n\n"
                "(optimiser steps of the reference's MNIST experiment model, tests/explib/mnist.yaml:44-77, on 65 536 resident rows: 15 eager\n"
                "steps in the trace -- 5 warm-up + 10 timed; per step 6 convolutions forward, 6 data-gradient convolutions and 6 weight\n"
                "gradients (usf_conv_wgrad_f32: conv_wgrad_kernel / conv_wgrad_k1_kernel + partial_sum_kernel), 2 pointwise convolutions each\n"
                "way, 5 channel-affine launches each way with their weight gradients, layer norm / gate forward and backward, SophiaG)\n\n"
                "| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|\n")
        for row in rows[:24]:
            name = row["Name"]
            name = name if len(name) < 110 else name[:107] + "..."
            f.write(f"| `{name}` | {row['Calls']} | {row['TotalDurationNs']} | {float(row['AverageNs']):.0f} | {row['Percentage']} |\n")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, "pmc_image_train_" + ctr)
        run(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--config", "mnist_image",
             "--mode", "train", "--eager-train", "--steps", "2", "--warmup", "1", "--no-kernel-timing"])
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                k = row["Kernel_Name"]
                if "usf::" not in k:
                    continue
                k = k[k.index("usf::") + 5:]
                k = k[: k.index("(")] if "(" in k else k
                agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kern = {}
    for k, dct in sorted(agg.items()):
        f_ = sum(dct["FETCH_SIZE"]) / max(1, len(dct["FETCH_SIZE"]))
        w_ = sum(dct["WRITE_SIZE"]) / max(1, len(dct["WRITE_SIZE"]))
        kern[k] = {"FETCH_SIZE_KB": round(f_, 1), "WRITE_SIZE_KB": round(w_, 1), "dispatches": len(dct["FETCH_SIZE"]),
                   "hbm_bytes_per_launch": int((2 * f_ + w_) * 1024)}
    json.dump({
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python3 bench.py --config mnist_image --mode train "
                  "--eager-train --steps 2 --warmup 1 --no-kernel-timing`, MI355X; tools/make_profiles.py",
        "units": "means per dispatch over ALL shapes a kernel instance served; hbm_bytes_per_launch = (2 FETCH_SIZE + WRITE_SIZE) KB "
                 "(FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Algorithmic bytes of a weight gradient at 65 536 rows "
                 "of 7 x 7 pixels = 4 (cin + cout) B per pixel: 32 -> 32: 822 MB, 16 -> 32 / 32 -> 16: 616 MB, 32 -> 64: 1233 MB, 16 -> 16: "
                 "411 MB (+ the per-wave partial sums: 1024 slots of 4 (256 tiles + 16) cout/16 cin/16 taps bytes); layer-norm backward: 12 B "
                 "per element = 1233 MB.",
        "kernels": kern}, open(os.path.join(out, f"{tag}_image_train_hbm_traffic.json"), "w"), indent=1)
    d = os.path.join(out, "pmc_image_train_mfma")
    run(["rocprofv3", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
         "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--config",
         "mnist_image", "--mode", "train", "--eager-train", "--steps", "2", "--warmup", "1", "--no-kernel-timing"])
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        mfma_util(fn, os.path.join(out, f"{tag}_image_train_mfma_util.json"))

# 7. the LIVE image configurations (round 4): radial base with an image-shaped loc, prior_scale 1.0, full depth
#    (experiments/mnist/mnist.yaml:44-92, fashion/fashionclasses_veriflow.yaml:55-93, cifar/cifar.yaml)
if want("live"):
    with open(os.path.join(out, f"{tag}_live_bench.jsonl"), "w") as f:
        for extra in (["--config", "mnist_live", "--steps", "5"], ["--config", "fashion_live", "--steps", "5"],
                      ["--config", "cifar_image", "--base", "radial", "--prior-scale", "1", "--steps", "5"],
                      ["--config", "mnist_image", "--base", "radial", "--prior-scale", "1", "--steps", "10"],
                      ["--config", "mnist_live", "--mode", "train", "--batch", "32", "--steps", "40", "--warmup", "6"],
                      ["--config", "fashion_live", "--mode", "train", "--batch", "32", "--steps", "40", "--warmup", "6"],
                      ["--config", "mnist_image", "--base", "radial", "--prior-scale", "1", "--mode", "train", "--batch", "32", "--steps", "50", "--warmup", "6"],
                      ["--config", "mnist_image", "--mode", "train", "--batch", "32", "--steps", "50", "--warmup", "6"],
                      ["--config", "mnist_image", "--base", "radial", "--prior-scale", "1", "--mode", "train", "--steps", "5"],
                      ["--config", "mnist_live", "--mode", "train", "--batch", "4096", "--steps", "5"]):
            r = run(["python3", "bench.py", "--cpu-seconds", "2", "--cpu-rows", "256"] + extra)
            f.write(last_json_line(r.stdout) + "\n")
    d = os.path.join(out, "ktrace_live")
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--config", "mnist_live",
         "--steps", "5", "--no-cpu-baseline"])
    stats = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, f"{tag}_live_kernel_stats.md"), "w") as f:
        f.write(f"# {tag}: rocprofv3 --kernel-trace --stats of `python3 bench.py --config mnist_live --steps 5 --no-cpu-baseline`\n\n"
                "(log_prob of 65 536 rows of the LIVE MNIST configuration, experiments/mnist/mnist.yaml:44-92: 15 coupling blocks x 3 gated\n"
                "layers, RadialDistribution(zeros[16,7,7], p=1, LogNormal(6, .35)) base on usf_radial_logprob_f32; 6 calls in the trace)\n\n"
                "| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|\n")
        for row in rows[:14]:
            name = row["Name"]
            name = name if len(name) < 110 else name[:107] + "..."
            f.write(f"| `{name}` | {row['Calls']} | {row['TotalDurationNs']} | {float(row['AverageNs']):.0f} | {row['Percentage']} |\n")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, "pmc_live_" + ctr)
        run(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", "python3", "bench.py", "--config", "mnist_image", "--base",
             "radial", "--prior-scale", "1", "--mode", "train", "--eager-train", "--steps", "2", "--warmup", "1", "--no-kernel-timing",
             "--no-cpu-baseline"])
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                k = row["Kernel_Name"]
                if "usf::radial" not in k:
                    continue
                k = k[k.index("usf::") + 5:]
                k = k[: k.index("(")] if "(" in k else k
                agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kern = {}
    for k, dct in sorted(agg.items()):
        f_ = sum(dct["FETCH_SIZE"]) / max(1, len(dct["FETCH_SIZE"]))
        w_ = sum(dct["WRITE_SIZE"]) / max(1, len(dct["WRITE_SIZE"]))
        kern[k] = {"FETCH_SIZE_KB": round(f_, 1), "WRITE_SIZE_KB": round(w_, 1), "dispatches": len(dct["FETCH_SIZE"]),
                   "hbm_bytes_per_launch": int((2 * f_ + w_) * 1024)}
    json.dump({
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python3 bench.py --config mnist_image --base radial "
                  "--prior-scale 1 --mode train --eager-train --steps 2 --warmup 1 --no-kernel-timing --no-cpu-baseline`, MI355X; "
                  "tools/make_profiles.py",
        "units": "the radial base's kernels at 65 536 rows x 784 features: hbm_bytes_per_launch = (2 FETCH_SIZE + WRITE_SIZE) KB (FETCH_SIZE "
                 "doubled as MI355X_MICROARCH.md prescribes for gfx950).  Algorithmic bytes: usf_radial_logprob_f32 reads the batch once "
                 "(4 D = 3136 B per sample: 205.5 MB) and writes 8 B per sample; usf_radial_logprob_grad_f32 reads it once and writes d/dz "
                 "(411 MB), the column sum for d/dloc reads d/dz once more (205.5 MB, colsum_kernel).",
        "kernels": kern}, open(os.path.join(out, f"{tag}_live_hbm_traffic.json"), "w"), indent=1)

print("wrote", sorted(os.listdir(out)))
