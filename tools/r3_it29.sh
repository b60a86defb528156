#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
rm -rf gpurun_out/r3/kt_small
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/kt_small -- python3 tools/image_small_batch.py ${1:-mnist_image} ${2:-100} > gpurun_out/r3/kt_small.log 2>&1
f=$(ls gpurun_out/r3/kt_small/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f'{r["Calls"]:>7} avg {float(r["AverageNs"])/1e3:>8.2f} us  {r["Name"][:110]}')
PY
grep rows gpurun_out/r3/kt_small.log
