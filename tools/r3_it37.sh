#!/bin/bash
# flat cfg2 log_prob at 100 rows (the reference's evaluation chunk): time and kernel trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python3 bench.py --batch 100 --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | cut -c1-260
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/kt_flat100 -- python3 bench.py --batch 100 --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/r3/kt_flat100.log 2>&1
f=$(ls gpurun_out/r3/kt_flat100/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("kernels", len(rows), "calls", sum(int(r["Calls"]) for r in rows), "total ms", sum(float(r["TotalDurationNs"]) for r in rows) / 1e6)
for r in rows[:14]:
    print(f'{r["Calls"]:>7} {float(r["TotalDurationNs"])/1e3:>10.1f} us  avg {float(r["AverageNs"])/1e3:>7.2f} us  {r["Name"][:120]}')
PY
