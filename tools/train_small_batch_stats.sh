# kernel time per training step of the cfg2 flow at the reference's default batch size (32 rows): is the step host- or GPU-bound?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
STEPS=20; WARM=5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train32_ktrace -- python3 bench.py --mode train --batch ${1:-32} --steps $STEPS --warmup $WARM --no-cpu-baseline --no-kernel-timing > gpurun_out/train32_rocprof.log 2>&1
f=$(find gpurun_out/train32_ktrace -name "*kernel_stats.csv" | head -1)
python3 - "$f" $STEPS $WARM <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) + int(sys.argv[3])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"all kernels: {tot / 1e6:.1f} ms over ~{steps} steps (+ set-up) = {tot / 1e6 / steps:.2f} ms/step, {calls / steps:.0f} launches/step")
for r in rows[:14]:
    n = r["Name"][:90]
    print(f"{n:92s} {r['Calls']:>7s} {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step  avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
tail -1 gpurun_out/train32_rocprof.log | cut -c1-300
