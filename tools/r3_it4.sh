#!/bin/bash
# kernel trace of Flow.fit (MNIST image model, batch 32, device training path): which kernels a replayed step consists of
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
FIT_IMAGE_ONLY=device rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/kt_fit32 -- python3 tools/fit_image.py ${1:-mnist_image} 32 > gpurun_out/r3/kt_fit32.log 2>&1
f=$(ls gpurun_out/r3/kt_fit32/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["Calls"]) for r in rows)
print("kernels", len(rows), "calls", tot)
for r in rows[:45]:
    print(f'{r["Calls"]:>7} {float(r["TotalDurationNs"])/1e3:>10.1f} us  avg {float(r["AverageNs"])/1e3:>7.2f} us  {r["Name"][:110]}')
PY
