#!/bin/bash
# round 3, experiment 2: coupling 32-row waves, fenced half-tile chunks
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp2.log
: > $O
for a in "65536 3 2" "32768 3 2" "65536 2 2" "65536 3 1" "1000 3 2" "65536 2 1"; do
  timeout -k 10 120 tools/exp_cplanes_x $a >> $O 2>&1 || echo "exp_cplanes_x $a rc $?" >> $O
done
timeout -k 10 120 tools/exp_cplanes_s 65536 3 2 >> $O 2>&1 || echo "rc $?" >> $O
timeout -k 10 120 tools/exp_cplanes_s 65536 2 2 >> $O 2>&1 || echo "rc $?" >> $O
echo "== fetch size per gemm variant" >> $O
cd /tmp && export TMPDIR=/tmp
for v in 0 3; do
  rm -rf /tmp/pmc$v
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc$v -o p -- $GRAFT_REPO_ROOT/tools/exp_planes_nt$v 65536 25 25 3 > /tmp/pmc$v.log 2>&1 || echo "pmc $v rc $?" >> $GRAFT_REPO_ROOT/$O
  find /tmp/pmc$v -type f | head -5 >> $GRAFT_REPO_ROOT/$O
  f=$(find /tmp/pmc$v -name "*counter_collection*.csv" | head -1)
  echo "-- nt$v $f" >> $GRAFT_REPO_ROOT/$O
  [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f >> $GRAFT_REPO_ROOT/$O 2>&1
  tail -3 /tmp/pmc$v.log >> $GRAFT_REPO_ROOT/$O
done
cat $GRAFT_REPO_ROOT/$O
