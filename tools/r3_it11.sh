#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/kt_wg -- python3 tools/bench_wgrad.py > gpurun_out/r3/kt_wg.log 2>&1
f=$(ls gpurun_out/r3/kt_wg/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(f'{r["Calls"]:>7} avg {float(r["AverageNs"])/1e3:>9.2f} us  min {float(r["MinNs"])/1e3:>9.2f}  {r["Name"][:100]}')
PY
