# kernel trace of the replayed Flow.fit step of the FLAT cfg2 model at the reference's training batch (bench.py --config cfg2 --mode fit
# --batch 32): launches per step and GPU time by kernel (gpurun_out/flat_fit32_stats.md) + the ordered trace of the last step
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
export TMPDIR=/tmp
STEPS=${FIT_STEPS:-30}; CFG=${FIT_CFG:-cfg2}; MODE=${FIT_MODE:-fit}
rm -rf gpurun_out/flat_ktrace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/flat_ktrace -- python3 bench.py --config $CFG --mode $MODE --batch 32 \
    --steps $STEPS --warmup 6 --no-cpu-baseline --no-kernel-timing --no-also > gpurun_out/flat_rocprof.log 2>&1
f=$(find gpurun_out/flat_ktrace -name "*kernel_stats.csv" | head -1)
t=$(find gpurun_out/flat_ktrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$t" "$STEPS" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[3])
with open("gpurun_out/flat_fit32_stats.md", "w") as o:
    o.write("| kernel | calls | per step | total ns | avg ns | % |\n|---|---|---|---|---|---|\n")
    for r in rows[:40]:
        n = r["Name"]; n = n if len(n) < 110 else n[:107] + "..."
        o.write(f"| `{n}` | {r['Calls']} | {int(r['Calls']) / (steps + 6):.1f} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['Percentage']} |\n")
tr = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
# the last replayed step: cut at the last sophiag step kernel
idx = [i for i, r in enumerate(tr) if "sophia" in r["Kernel_Name"].lower()]
if len(idx) >= 2:
    # steps end with the optimiser kernels: find the last two groups
    ends = [i for k, i in enumerate(idx) if k + 1 == len(idx) or idx[k + 1] - i > 5]
    a, b = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(tr))
    with open("gpurun_out/flat_fit32_trace.txt", "w") as o:
        t0 = int(tr[a]["Start_Timestamp"]); prev = t0
        for r in tr[a:b]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            o.write(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev) / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'][:120]}\n")
            prev = e
        o.write(f"launches {b - a}, span {(int(tr[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us\n")
print(open("gpurun_out/flat_fit32_stats.md").read())
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/flat_rocprof.log | head -2
tail -1 gpurun_out/flat_fit32_trace.txt
