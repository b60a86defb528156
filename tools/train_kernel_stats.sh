#!/bin/bash
# rocprofv3 --kernel-trace --stats of the flat training step (cfg2, 65 536 rows): per-kernel averages -> gpurun_out/<tag>_train_stats.csv
#   tools/train_kernel_stats.sh <tag> [extra bench.py flags]
tag=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/prof_train_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --mode train --steps 4 --warmup 2 --no-also --no-cpu-baseline "$@" > $R/gpurun_out/prof_train_$tag.log 2>&1 || exit 1
f=$(find $d -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:22]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:9.1f} us total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
