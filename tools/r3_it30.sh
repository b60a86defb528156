#!/bin/bash
# kernel trace of Flow.fit (cfg2 flat flow, batch 32, replayed step): which kernels a step consists of
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
FIT_ONLY=graph rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/kt_flat32 -- python3 tools/fit_small_batch.py 32 40 > gpurun_out/r3/kt_flat32.log 2>&1
tail -2 gpurun_out/r3/kt_flat32.log
f=$(ls gpurun_out/r3/kt_flat32/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["Calls"]) for r in rows)
print("kernels", len(rows), "calls", tot, "total ms", sum(float(r["TotalDurationNs"]) for r in rows) / 1e6)
for r in rows[:50]:
    print(f'{r["Calls"]:>7} {float(r["TotalDurationNs"])/1e3:>10.1f} us  avg {float(r["AverageNs"])/1e3:>7.2f} us  {r["Name"][:120]}')
PY
