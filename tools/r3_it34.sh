#!/bin/bash
# skinny GEMM variants at 32 / 100 rows (tools/bench_skinny.py)
cd $GRAFT_REPO_ROOT
for v in "4 4" "8 4" "8 8" "4 8"; do
  set -- $v
  USF_SKINNY_KS_SMALL=$1 USF_SKINNY_G=$2 python3 tools/bench_skinny.py 32 2>&1 | grep rows
done
USF_SKINNY_G=4 python3 tools/bench_skinny.py 100 2>&1 | grep rows
USF_SKINNY_G=8 python3 tools/bench_skinny.py 100 2>&1 | grep rows
