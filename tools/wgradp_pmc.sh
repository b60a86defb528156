#!/bin/bash
# L2 / fabric counters of the planes weight-gradient kernel in the harness (tools/exp_wgradp): tools/wgradp_pmc.sh [M N K]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  d=$R/gpurun_out/wgp_pmc_$(echo $c | tr ' ' '_')
  rm -rf $d
  timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d $d -- $R/tools/exp_wgradp ${1:-65536} ${2:-784} ${3:-784} > $R/gpurun_out/wgp_pmc.log 2>&1 || exit 1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_summary.py $f | grep -A3 "wgrad_planes"
done
