#!/bin/bash
# round 3, experiment 1: coupling 16-row vs 32-row waves (bit-compare + timing + stamps); planes GEMM cache-policy variants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp1.log
: > $O
echo "== coupling" >> $O
timeout -k 10 120 tools/exp_cplanes_x 65536 3 2 >> $O 2>&1 || echo "exp_cplanes_x rc $?" >> $O
timeout -k 10 120 tools/exp_cplanes_x 32768 3 2 >> $O 2>&1 || echo "rc $?" >> $O
timeout -k 10 120 tools/exp_cplanes_x 65536 2 2 >> $O 2>&1 || echo "rc $?" >> $O
timeout -k 10 120 tools/exp_cplanes_x 65536 3 1 >> $O 2>&1 || echo "rc $?" >> $O
timeout -k 10 120 tools/exp_cplanes_x 1000 3 2 >> $O 2>&1 || echo "rc $?" >> $O
timeout -k 10 120 tools/exp_cplanes_s 65536 3 2 >> $O 2>&1 || echo "rc $?" >> $O
echo "== gemm nt variants (two rounds)" >> $O
for r in 1 2; do for v in 0 1 2 3; do echo "-- nt$v" >> $O; timeout -k 10 120 tools/exp_planes_nt$v 65536 25 25 3 >> $O 2>&1 || echo "rc $?" >> $O; done; done
echo "== fetch size per variant" >> $O
cd /tmp && export TMPDIR=/tmp
for v in 0 1 3; do
  rm -rf /tmp/pmc$v
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc$v -o p -- $GRAFT_REPO_ROOT/tools/exp_planes_nt$v 65536 25 25 3 > /tmp/pmc$v.log 2>&1 || echo "pmc $v rc $?" >> $GRAFT_REPO_ROOT/$O
  f=$(find /tmp/pmc$v -name "*counter_collection.csv" | head -1)
  echo "-- nt$v $f" >> $GRAFT_REPO_ROOT/$O
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f >> $GRAFT_REPO_ROOT/$O 2>&1
done
cat $GRAFT_REPO_ROOT/$O
