# kernel trace of the replayed Flow.fit step of the LIVE MNIST configuration (bench.py --config mnist_live --mode train) at a
# small batch: launches per step and GPU time per step by kernel (gpurun_out/live_small_batch_stats.md)
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
export TMPDIR=/tmp
B=${LIVE_BATCH:-32}; STEPS=${LIVE_STEPS:-40}; CFG=${LIVE_CFG:-mnist_live}
rm -rf gpurun_out/live_ktrace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/live_ktrace -- python3 bench.py --config $CFG --mode train --batch $B \
    --steps $STEPS --warmup 6 --no-cpu-baseline --no-kernel-timing > gpurun_out/live_rocprof.log 2>&1
f=$(find gpurun_out/live_ktrace -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$STEPS" "$B" "$CFG" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps, B, cfg = int(sys.argv[2]), sys.argv[3], sys.argv[4]
tot_calls = sum(int(r["Calls"]) for r in rows)
tot_ns = sum(int(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/live_small_batch_stats.md", "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats of `bench.py --config {cfg} --mode train --batch {B} --steps {steps} --warmup 6`\n\n"
            f"(6 warm-up steps incl. 3 eager ones and the capture, then {steps} replayed steps; all kernels of the process: {tot_calls} launches, "
            f"{tot_ns / 1e6:.1f} ms of GPU time)\n\n| kernel | calls | total ns | avg ns | % |\n|---|---|---|---|---|\n")
    for r in rows[:25]:
        n = r["Name"]; n = n if len(n) < 100 else n[:97] + "..."
        o.write(f"| `{n}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['Percentage']} |\n")
print(open("gpurun_out/live_small_batch_stats.md").read())
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/live_rocprof.log | head -2
